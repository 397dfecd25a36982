#!/bin/bash
# usage: tools/probe/isa.sh <file.hip> <mangled-substring> [first-mfma-context-lines]
# Compiles one csrc file to gfx950 ISA and prints the lines around the first MFMA of the chosen kernel (comment lines stripped).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
F=$1; SUB=$2; N=${3:-80}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/zs3_amd/csrc -Wno-unused-result \
  --cuda-device-only -S $ROOT/zs3_amd/csrc/$F -o /tmp/isa_out.s 2>&1 | grep -iv "warning" || true
python3 - "$SUB" "$N" <<'PY'
import sys,re
s=open('/tmp/isa_out.s').read()
sub,n=sys.argv[1],int(sys.argv[2])
m=re.search(r'^(\S*'+re.escape(sub)+r'[^:\s]*):', s, re.M)
i=m.start(); j=s.index('s_endpgm',i)
k=[l for l in s[i:j].split('\n') if not l.strip().startswith(';')]
open('/tmp/isa_kernel.s','w').write('\n'.join(k))
idx=[q for q,l in enumerate(k) if 'v_mfma' in l]
print(m.group(1), 'lines', len(k), 'mfma', len(idx))
for q in range(max(0,idx[0]-3), min(len(k), idx[0]+n)): print(k[q][:110])
PY
