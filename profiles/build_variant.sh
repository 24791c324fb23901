#!/bin/bash
# Build the library of an OLDER git revision as bsms-gnn_amd/lib_<name>.so.keep for a same-box A/B (profiles/ab.sh):
#   bash profiles/build_variant.sh <git-rev> <name>
# csrc/ and include/ are taken from <git-rev>; host-only files whose ABI grew since (hierarchy.hip + the header) are
# taken from the working tree so that today's Python binding still loads the old kernels.
set -e
rev=$1; name=$2
root="$(cd "$(dirname "$0")/.." && pwd)"
tmp=$(mktemp -d)
mkdir -p $tmp/bsms-gnn_amd/csrc $tmp/include
for f in $(git -C $root ls-tree --name-only $rev bsms-gnn_amd/csrc/); do git -C $root show $rev:$f > $tmp/$f; done
cp $root/include/bsms_hip.h $tmp/include/
cp $root/bsms-gnn_amd/csrc/hierarchy.hip $tmp/bsms-gnn_amd/csrc/
[ -f $tmp/bsms-gnn_amd/csrc/sim.hip ] || cp $root/bsms-gnn_amd/csrc/sim.hip $tmp/bsms-gnn_amd/csrc/
cd $tmp/bsms-gnn_amd/csrc
objs=""
for s in $(ls *.hip | sed s/.hip//); do
  extra=""; { [ $s = rowsum ] || [ $s = sim ]; } && extra="-ffp-contract=off"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $extra -c $s.hip -o $s.o &
  objs="$objs $s.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/bsms-gnn_amd/lib_$name.so.keep $objs
rm -rf $tmp
echo "built lib_$name.so.keep from $rev"
