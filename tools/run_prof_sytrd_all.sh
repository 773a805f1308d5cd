# numbers behind profiles/r02_eigh_sytrd.txt
echo "== tools/probe_sytrd.py: clo_sytrd_f32 + sstedc + sormtr vs torch.linalg.eigh (single matrix, KFAC-like rank-513 PSD)"
python tools/probe_sytrd.py 333 577 1153 2305 4609 7001 2>&1 | grep "^n="
echo "== tools/probe_rocsolver_phases.py: where rocSOLVER spends its time"
python tools/probe_rocsolver_phases.py 577 1153 2305 4609 2>&1 | grep "^n="
echo "== tools/probe_sytrd_phases.py: cost of the phases of the column launch (CLO_TD_DEBUG: 32 = empty kernels, 16 = prologue only, 17 = prologue without the partial sums, 2 = no row loop, 0 = everything)"
for d in 32 17 16 2 0; do CLO_TD_DEBUG=$d python tools/probe_sytrd_phases.py 577 2305 4609 2>&1 | grep dbg; done
echo "== tools/probe_sytrd_concurrent.py: several matrices at once"
python tools/probe_sytrd_concurrent.py 4609 3 2>&1 | grep "n="
python tools/probe_sytrd_concurrent.py 1153 4 2>&1 | grep "n="
