# usage: buildone.sh name file.hip "-DFLAGS"  -> curvlinops_amd/lib/variants/libclo_<name>.so: the default library with only
# csrc/<file> rebuilt with the flags (the other objects are those of the in-tree build; load with CLO_HIP_LIB=<path>)
set -e
cd /root/repo/curvlinops_amd/csrc
name=$1; file=$2; shift; shift
(cd /root/repo && python -c "from curvlinops_amd.csrc.build import build; build()")
mkdir -p ../lib/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $file -o /tmp/one_$name.o
base=${file%.hip}
objs=$(ls ../lib/obj/*.o | grep -v "/$base.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libclo_$name.so $objs /tmp/one_$name.o
echo built $name
