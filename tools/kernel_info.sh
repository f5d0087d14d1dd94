#!/bin/bash
# Register / scratch / LDS usage of the device kernels in an object or shared library built by hipcc:
#   tools/kernel_info.sh pysph_amd/csrc/sph_eval.o [name-regex]      (regex on the demangled kernel name)
# and, with DISASM=1, the ISA of the first matching kernel into /tmp/kernel.s
set -e
LLVM=/opt/rocm/lib/llvm/bin
OBJ=$1; PAT=${2:-.}
TMP=$(mktemp -d)
$LLVM/clang-offload-bundler --type=o --input=$OBJ --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/dev.co --unbundle 2>/dev/null || \
  { $LLVM/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $OBJ; $LLVM/clang-offload-bundler --type=o --input=$TMP/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/dev.co --unbundle; }
$LLVM/llvm-readelf --notes $TMP/dev.co > $TMP/notes.txt
python3 - "$TMP/notes.txt" "$PAT" <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
ks = re.split(r'\n\s+- \.agpr_count:', txt)
rows = []
for k in ks[1:]:
    g = lambda key: (re.search(r'\.%s:\s+(\S+)' % key, k) or [None, '?'])[1]
    name = g('name')
    try:
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    if pat.search(dem):
        rows.append((dem[:110], g('vgpr_count'), g('sgpr_count'), g('sgpr_spill_count'), g('vgpr_spill_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
print('%-110s %5s %5s %6s %6s %7s %6s' % ('kernel', 'vgpr', 'sgpr', 'sspill', 'vspill', 'scratch', 'lds'))
for r in rows:
    print('%-110s %5s %5s %6s %6s %7s %6s' % r)
PY
if [ -n "${DISASM:-}" ]; then
  $LLVM/llvm-objdump -d --no-show-raw-insn $TMP/dev.co > /tmp/all_kernels.s
  echo "ISA of all kernels: /tmp/all_kernels.s"
fi
rm -rf $TMP
