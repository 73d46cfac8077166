#!/bin/bash
# usage: mkvar.sh name flags...
name=$1; shift
cd /root/repo/semi-supervised-asr_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -Wno-inline-asm -o /root/repo/scratchlibs/$name.so *.hip 2>&1 | grep -v warning | grep -i "error" 
echo built $name
