#!/bin/bash
# randomised parity campaigns on the GPU (tools/parity_campaign.py), four processes side by side: the library's own kernel choice and the
# lanes = reads kernel forced, with and without TraceItem lists.  Results -> gpurun_out/campaign_*.json
cd "${GRAFT_REPO_ROOT:-.}"
T=${1:-210}
timeout -k 10 1000 python tools/parity_campaign.py --trials $T --seed 3101 > gpurun_out/campaign_default_trace.json 2> gpurun_out/campaign_a.err &
P1=$!
GA_LANES=1 timeout -k 10 1000 python tools/parity_campaign.py --trials $T --seed 3102 > gpurun_out/campaign_lanes_trace.json 2> gpurun_out/campaign_b.err &
P2=$!
timeout -k 10 1000 python tools/parity_campaign.py --trials $T --seed 3103 --no-trace > gpurun_out/campaign_default_notrace.json 2> gpurun_out/campaign_c.err &
P3=$!
GA_LANES=1 timeout -k 10 1000 python tools/parity_campaign.py --trials $T --seed 3104 --no-trace > gpurun_out/campaign_lanes_notrace.json 2> gpurun_out/campaign_d.err &
P4=$!
wait $P1 $P2 $P3 $P4
cat gpurun_out/campaign_default_trace.json gpurun_out/campaign_lanes_trace.json gpurun_out/campaign_default_notrace.json gpurun_out/campaign_lanes_notrace.json
