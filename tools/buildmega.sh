# usage: buildmega.sh name "-DFLAGS"  -> curvlinops_amd/lib/variants/libclo_<name>.so: the default library with only
# csrc/mlp_mega.hip rebuilt with the flags (the other objects are shared; load with CLO_HIP_LIB=<path>)
set -e
cd /root/repo/curvlinops_amd/csrc
name=$1; shift
mkdir -p /tmp/obj_base ../lib/variants
for f in gemm gemm_v3 mlp stream_ops linalg conv gram sytrd eigh eigh_driver; do
  if [ ! -f /tmp/obj_base/$f.o ] || [ $f.hip -nt /tmp/obj_base/$f.o ]; then
    ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f.hip -o /tmp/obj_base/$f.o ) &
  fi
done
wait
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c mlp_mega.hip -o /tmp/obj_base/mega_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libclo_$name.so /tmp/obj_base/{gemm,gemm_v3,mlp,stream_ops,linalg,conv,gram,sytrd,eigh,eigh_driver}.o /tmp/obj_base/mega_$name.o
echo built $name
