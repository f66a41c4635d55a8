#!/bin/bash
# does every committed profiles/r04_*_pmc.json still describe the code object bench.py runs?  (bench.py --no-live-pmc quotes a summary only when kernel id and workload match)
for s in "--workload C3" "--workload C4" "--workload C5" "--scene hexagons --size 4096 --height 2048" "--scene mesh --size 2048" "--scene here_be_dragons --size 1000 --height 400" "--scene reflect_refract --size 4096 --height 2048" "--scene first_textures --size 4096 --height 2048"; do
  python3 bench.py $s --no-live-pmc --steps 3 --warmup 2 --cpu-seconds 0 --no-one-shot --no-verify 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read())['roofline']
print('%-60s %-36s %s' % ('$s', r['kernel_id'], r['pmc_source']))"
done
