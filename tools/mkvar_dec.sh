#!/bin/bash
# Measurement builds that differ in csrc/dec_persist.hip only: the other sources are compiled once, each variant compiles
# dec_persist.hip with its -D flags and links.   usage: mkvar_dec.sh name1 "flags1" name2 "flags2" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; C=$R/semi-supervised-asr_amd/csrc; O=$R/scratchlibs/obj; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm"
for s in $C/*.hip; do b=$(basename $s .hip); [ $b = dec_persist ] && continue; ( [ -f $O/$b.o ] || /opt/rocm/bin/hipcc $F -c $s -o $O/$b.o 2>&1 | grep -i " error" ) & done; wait
while [ $# -ge 2 ]; do
  ( /opt/rocm/bin/hipcc $F $2 -c $C/dec_persist.hip -o $O/dec_$1.o 2>&1 | grep -i " error"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $R/scratchlibs/$1.so $O/dec_$1.o $(ls $O/*.o | grep -v "/dec_") && echo built $1 ) &
  shift 2
done; wait
