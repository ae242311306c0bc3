#!/bin/bash
# A/B runs of the benchmark step under different library options (environment form: PSVAE_<OPTION NAME>=value).
# Usage: tools/ab.sh <weak|strong|strong:GLOBAL_BATCH> "VAR=1 VAR2=0" "..." ...   (a setting like "A=0" = defaults)
# Prints molecules/s, ms/step and host enqueue time per setting, two runs each.
mode=$1; shift
O=gpurun_out; mkdir -p $O
for cfg in "$@"; do
  for rep in 1 2; do
    case $mode in
      strong:*) A="--scaling strong --global-batch ${mode#strong:} --steps 60 --warmup 10";;
      strong) A="--scaling strong --steps 40 --warmup 8";;
      *) A="--steps 200 --warmup 20";;
    esac
    r=$(env $cfg timeout -k 10 120 python3 bench.py $A --no-cpu-baseline --no-bulk --no-configs --config3-steps 0 --profile-steps 1 2>$O/ab.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f mol/s %.4f ms host %.3f' % (d['value'], d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step', -1)))") || { echo "FAILED: $cfg"; tail -5 $O/ab.err; exit 1; }
    echo "[$mode] ${cfg:-default}: $r"
  done
done
