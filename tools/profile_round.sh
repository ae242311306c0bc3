#!/bin/bash
# One measurement pass on the GPU box: PMC traffic and MFMA passes, the rocprofv3 kernel statistics of the
# bench command, then the bench line itself (which quotes the traffic file).  Usage: tools/profile_round.sh <tag> [round]
# Everything lands under gpurun_out/ (the step's traffic file also under profiles/<round>_pmc_traffic.json, which bench.py
# quotes); copy what else is to be kept into profiles/.
set -e -o pipefail
V=${1:-vX}
R=${2:-r03}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
PMCARGS="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bulk --no-configs --profile-steps 1 --config3-steps 0"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_${V}_f -o f --output-format csv -- python3 $PMCARGS > $O/pmc_${V}_f.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_${V}_w -o w --output-format csv -- python3 $PMCARGS > $O/pmc_${V}_w.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_${V}_m -o m --output-format csv -- python3 $PMCARGS > $O/pmc_${V}_m.log 2>&1
F=$(find $O/pmc_${V}_f -name '*counter_collection.csv' | head -1)
W=$(find $O/pmc_${V}_w -name '*counter_collection.csv' | head -1)
M=$(find $O/pmc_${V}_m -name '*counter_collection.csv' | head -1)
python3 tools/pmc_traffic.py $F $W $O/pmc_traffic_${V}.json > $O/pmc_traffic_${V}.txt
(cd tools && python3 pmc_mfma.py ../$M ../$O/pmc_mfma_${V}.json) > $O/pmc_mfma_${V}.txt
cp $O/pmc_traffic_${V}.json profiles/${R}_pmc_traffic.json
# the message-passing kernels at bulk size (4096 graphs per launch, HBM-resident rotating buffers)
BULK="tools/agg_bench.py 4096 --iters=20"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_${V}_bf -o f --output-format csv -- python3 $BULK > $O/pmc_${V}_bf.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_${V}_bw -o w --output-format csv -- python3 $BULK > $O/pmc_${V}_bw.log 2>&1
python3 tools/pmc_traffic.py $(find $O/pmc_${V}_bf -name '*counter_collection.csv' | head -1) $(find $O/pmc_${V}_bw -name '*counter_collection.csv' | head -1) \
    $O/pmc_bulk_traffic_${V}.json "$BULK" "4096 ZINC-shape graphs per launch (95140 atoms, 205072 directed bonds, hidden 300), rotating buffers" > $O/pmc_bulk_traffic_${V}.txt
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/prof_${V} -o ${V} --output-format csv -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-bulk --no-configs --config3-steps 0 > $O/prof_${V}.log 2>&1
timeout -k 10 400 python3 bench.py > $O/bench_${V}.json 2> $O/bench_${V}.err
python3 -c "import json;d=json.load(open('$O/bench_${V}.json'));print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline']['value'])"
