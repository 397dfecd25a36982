"""stdin: a bench.py run's stdout (RCCL prints its banner there too) -> the chosen fields of its JSON line.  python tools/probe/jline.py tag key[.sub] ..."""
import json
import sys

d = json.loads([ln for ln in sys.stdin.read().splitlines() if ln.startswith("{")][-1])
out = []
for k in sys.argv[2:]:
    v = d
    for part in k.split("."):
        v = v.get(part) if isinstance(v, dict) else None
    out.append(f"{k}={round(v, 3) if isinstance(v, float) else v}")
print(sys.argv[1], *out)
