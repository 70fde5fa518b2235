#!/bin/bash
# tools/isa_of.sh <mangled-substring>: compiles csrc/conv.hip with -save-temps and prints the instruction skeleton (loads, stores, waits,
# MFMAs) and the register numbers of the first kernel whose mangled name contains the substring.  A development aid, not part of the build.
set -e
C=/root/repo/simclr_amd/csrc
cd $C
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -c conv.hip -o build/conv.o -save-temps=obj
S=$C/build/conv-hip-amdgcn-amd-amdhsa-gfx950.s
N=$(grep -o "^_ZN[A-Za-z0-9_]*$1[A-Za-z0-9_]*:" $S | head -1 | tr -d ':')
echo "kernel $N"
awk -v n="$N" '$0 ~ "^"n":" {f=1} f {print} f && /\.end_amdhsa_kernel/ {exit}' $S > /tmp/isa_of.s
grep -E "next_free_vgpr|private_segment_fixed|accum_offset" /tmp/isa_of.s
grep -n -E "s_waitcnt vmcnt|global_load|global_store|v_mfma|s_barrier" /tmp/isa_of.s | awk '{print $2, $3}' | sed -e 's/global_load_dwordx4.*/LD/' -e 's/v_mfma[a-z0-9_]*.*/M/' -e 's/global_store_dwordx4.*/ST/' -e 's/s_waitcnt vmcnt/W/' | tr '\n' ' ' | sed -e 's/ , / /g' | head -c 6000
echo
rm -f $C/build/conv-h*.bc $C/build/conv-h*.hipi $C/build/conv-h*.out* $C/build/conv-host* $C/build/conv.hip-hip-*.hipfb $C/build/conv-hip-*.o
