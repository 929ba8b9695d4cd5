#!/bin/bash
# development: disassemble a kernel of et_forward_tile.o into /tmp/dis/<name>.s    usage: disasm_ws.sh [mangled-name-fragment]
mkdir -p /tmp/dis && cd /tmp/dis || exit 1
OBJ=${OBJ:-/root/repo/epipolar_transformers_amd/lib/obj/et_forward_tile.o}
FRAG=${1:-tile_ws_kernelILi256E}
cp "$OBJ" ft.o
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading ft.o > /dev/null 2>&1
mv ft.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 ft.co; rm -f ft.o.0.host*
/opt/rocm/lib/llvm/bin/llvm-objdump -d ft.co > ft.s
a=$(grep -n "^[0-9a-f]* <.*$FRAG" ft.s | head -1 | cut -d: -f1)
b=$(grep -n "^[0-9a-f]* <" ft.s | awk -F: -v a="$a" '$1>a{print $1; exit}')
[ -z "$b" ] && b=$(wc -l < ft.s)
sed -n "${a},${b}p" ft.s > k.s
echo "lines $(wc -l < k.s)  v_mfma $(grep -c v_mfma k.s)  barriers at: $(grep -n s_barrier k.s | cut -d: -f1 | tr '\n' ' ')"
