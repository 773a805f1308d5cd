# usage: buildmega.sh name "-DFLAGS"  -> curvlinops_amd/lib/variants/libclo_<name>.so: the default library with only
# csrc/mlp_mega.hip rebuilt with the flags (the other objects are those of the in-tree build, curvlinops_amd/lib/obj;
# load the variant with CLO_HIP_LIB=<path>)
set -e
cd /root/repo/curvlinops_amd/csrc
name=$1; shift
(cd /root/repo && python -c "from curvlinops_amd.csrc.build import build; build()")
mkdir -p ../lib/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c mlp_mega.hip -o /tmp/mega_$name.o
objs=$(ls ../lib/obj/*.o | grep -v mlp_mega.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libclo_$name.so $objs /tmp/mega_$name.o
echo built $name
