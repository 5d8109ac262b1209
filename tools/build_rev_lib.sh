#!/bin/bash
# Build the kernels of another git revision (default HEAD) into ab/libtcct_REV.so for same-box A/B timing:
#   bash tools/build_rev_lib.sh [REV]   ->  TCCT_LIB_PATH=ab/libtcct_REV.so python tools/kbench.py ...
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p $TMP/tcct_amd/csrc $TMP/include $ROOT/ab
for f in $(git -C $ROOT ls-tree --name-only $REV tcct_amd/csrc/ | grep -E '\.(hip|h|inc)$'); do git -C $ROOT show $REV:$f > $TMP/$f; done
git -C $ROOT show $REV:include/tcct_hip.h > $TMP/include/tcct_hip.h
cd $TMP/tcct_amd/csrc
ls *.hip | xargs -P 4 -I{} /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c {} -o {}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab/libtcct_$(git -C $ROOT rev-parse --short $REV).so *.o
rm -rf $TMP
ls -la $ROOT/ab
