#!/bin/bash
# same-box A/B of bench.py under different environment settings (one JSON summary line each)
# usage: bash tools/ab_bench.sh "FT_RNN_LOCAL=1" "FT_RNN_LOCAL=0" ...
for cfg in "$@"; do
  env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --family-steps 2 > /tmp/ab.json 2> /tmp/ab.err || { tail -5 /tmp/ab.err; continue; }
  python - "$cfg" <<'PY'
import json, sys
d = json.load(open('/tmp/ab.json'))
fam = d['roofline']['families']
print(sys.argv[1], '| ms/step', d['ms_per_step'], '| rnn', d['rnn_launches'], '| bank fwd ms', d['roofline']['launch_ms'],
      '| trunk rnn ms', fam[0]['ms_per_step'] if fam else None, '| pred rnn ms', fam[1]['ms_per_step'] if len(fam) > 1 else None, flush=True)
PY
done
