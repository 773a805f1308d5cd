# usage: buildvar.sh name "-DFLAGS"   -> curvlinops_amd/lib/variants/libclo_<name>.so (load with CLO_HIP_LIB=<path>)
set -e
cd /root/repo/curvlinops_amd/csrc
name=$1; shift
mkdir -p /tmp/obj_$name ../lib/variants
for f in gemm gemm_v3 mlp mlp_mega stream_ops linalg conv gram syrk_grouped sytrd eigh eigh_driver kron; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $f.hip -o /tmp/obj_$name/$f.o ) &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libclo_$name.so /tmp/obj_$name/*.o
echo built $name
