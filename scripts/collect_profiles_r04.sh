#!/bin/bash
# copies the summaries of scripts/profile_r04.sh (gpurun_out/prof_r04/) into profiles/ under their round-4 names and rebuilds profiles/traffic.json
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/prof_r04; D=profiles
for c in cfg1 cfg2 cfg3 cfg4 cfg5 inp cfg2_k1 dropin dropin_flat; do [ -f $S/bench_$c.json ] && cp $S/bench_$c.json $D/r04_bench_$c.json; done
cp $S/kernel_stats_cfg2.csv $D/r04_kernel_stats.csv
for c in cfg1 cfg3 cfg4 cfg5 inp cfg2_k1; do cp $S/kernel_stats_$c.csv $D/r04_${c}_kernel_stats.csv; done
cp $S/timeline.txt $D/r04_timeline.txt; cp $S/kernel_table_cfg2.txt $D/r04_kernel_table_cfg2.txt
cp $S/pmc_hbm_traffic.txt $D/r04_pmc_hbm_traffic.txt; cp $S/layer_traffic.txt $D/r04_layer_traffic.txt
cp $S/pmc_sq_up9.txt $D/r04_pmc_sq_up9.txt; cp $S/pmc_sq_up7.txt $D/r04_pmc_sq_up7.txt
cp $S/bwdx6_layers.txt $D/r04_bwdx6_layers.txt; cp $S/bwdx6s_prof.txt $D/r04_bwdx6s_prof.txt; cp $S/bwdx6s_variants.txt $D/r04_bwdx6s_variants.txt; cp $S/x6_layers.txt $D/r04_x6_layers.txt; cp $S/graph_bench.txt $D/r04_graph_bench_final.txt
python3 scripts/make_traffic_json.py $D/r04_layer_traffic.txt mfvi-dip-mia_amd/libmfvi_hip.so > $D/traffic.json
echo "profiles/ updated; traffic.json stamped with $(python3 -c "import json;print(json.load(open('profiles/traffic.json'))['csrc_sha256'][:16])")"
