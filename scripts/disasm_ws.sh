#!/bin/bash
# development: disassemble the warp-specialised forward kernel (NV8) into /tmp/dis/ws8.s and list its vmcnt waits
mkdir -p /tmp/dis && cd /tmp/dis || exit 1
OBJ=/root/repo/epipolar_transformers_amd/lib/obj/et_forward_tile.o
cp "$OBJ" ft.o
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading ft.o > /dev/null 2>&1
mv ft.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 ft.co; rm -f ft.o.0.host*
/opt/rocm/lib/llvm/bin/llvm-objdump -d ft.co > ft.s
a=$(grep -n "ws_kernelILi256ELi8" ft.s | head -1 | cut -d: -f1)
b=$(grep -n "^[0-9a-f]* <" ft.s | awk -F: -v a="$a" '$1>a{print $1; exit}')
sed -n "${a},${b}p" ft.s > ws8.s
grep -n "vmcnt" ws8.s | sed 's/\/\/.*//' | awk '{printf "%s %s %s | ", $1, $3, $4; if (NR%6==0) print ""}'; echo
grep -c v_mfma ws8.s
