#!/bin/bash
# Bench lines of the widened model families (SURVEY.md §8 f): usage tools/gpu_evidence_models.sh <tag>
tag=${1:-r01}
out=$PWD/gpurun_out
mkdir -p "$out"
python3 bench.py --model qwen > "$out/${tag}_qwen_bench.json" 2> "$out/${tag}_qwen.err"
python3 bench.py --model qwen --batch-cfg > "$out/${tag}_qwen_batchcfg_bench.json" 2> "$out/${tag}_qwen_batchcfg.err"
python3 bench.py --quant bf16 --height 1024 --width 1024 > "$out/${tag}_flux_bf16_1024_bench.json" 2> "$out/${tag}_flux_bf16_1024.err"
python3 bench.py --quant int8 > "$out/${tag}_flux_int8_bench.json" 2> "$out/${tag}_flux_int8.err"
python3 bench.py --model sd35 > "$out/${tag}_sd35_bench.json" 2> "$out/${tag}_sd35.err"
python3 bench.py --model sdxl --nchw --graph > "$out/${tag}_sdxl_nchw_graph_bench.json" 2> "$out/${tag}_sdxl_nchw.err"
python3 bench.py --model sdxl > "$out/${tag}_sdxl_bench.json" 2> "$out/${tag}_sdxl.err"
python3 bench.py --model sdxl --graph > "$out/${tag}_sdxl_graph_bench.json" 2> "$out/${tag}_sdxl_graph.err"
# BASELINE.json configs[0]: SDXL-base UNet 512 x 512, 4 steps (the line carries the CPU baseline of the same layers)
python3 bench.py --model sdxl --height 512 --width 512 --steps 4 --warmup 1 > "$out/${tag}_sdxl_512_bench.json" 2> "$out/${tag}_sdxl_512.err"
python3 bench.py --model wan > "$out/${tag}_wan_dense_bench.json" 2> "$out/${tag}_wan_dense.err"
python3 bench.py --model wan --start-step 11 --sparse-config examples/sparse/radial_attn_wan.json > "$out/${tag}_wan_sparse_bench.json" 2> "$out/${tag}_wan_sparse.err"
# OPT-IN quantised attention (the reference's SageAttention / SpargeAttn class; never the headline): the same workloads, other attention arithmetic
python3 bench.py --attn-pv-fp8 --no-cpu-baseline > "$out/${tag}_flux_pvfp8_bench.json" 2> "$out/${tag}_flux_pvfp8.err"
python3 bench.py --attn-fp8 --no-cpu-baseline > "$out/${tag}_flux_attnfp8_bench.json" 2> "$out/${tag}_flux_attnfp8.err"
python3 bench.py --model qwen --attn-fp8 --no-cpu-baseline > "$out/${tag}_qwen_attnfp8_bench.json" 2> "$out/${tag}_qwen_attnfp8.err"
python3 bench.py --model qwen --batch-cfg --attn-fp8 --no-cpu-baseline > "$out/${tag}_qwen_batchcfg_attnfp8_bench.json" 2> "$out/${tag}_qwen_batchcfg_attnfp8.err"
python3 bench.py --model wan --attn-fp8 --no-cpu-baseline > "$out/${tag}_wan_dense_attnfp8_bench.json" 2> "$out/${tag}_wan_dense_attnfp8.err"
python3 bench.py --model wan --start-step 11 --sparse-config examples/sparse/radial_attn_wan.json --attn-pv-fp8 --no-cpu-baseline > "$out/${tag}_wan_sparse_pvfp8_bench.json" 2> "$out/${tag}_wan_sparse_pvfp8.err"
python3 bench.py --model wan --start-step 11 --sparse-config examples/sparse/radial_attn_wan.json --attn-fp8 --no-cpu-baseline > "$out/${tag}_wan_sparse_attnfp8_bench.json" 2> "$out/${tag}_wan_sparse_attnfp8.err"
for f in flux_pvfp8 flux_attnfp8 qwen_attnfp8 qwen_batchcfg_attnfp8 wan_dense_attnfp8 wan_sparse_pvfp8 wan_sparse_attnfp8 qwen qwen_batchcfg flux_bf16_1024 flux_int8 sd35 sdxl sdxl_graph sdxl_512 wan_dense wan_sparse; do echo "$f: $(tail -1 $out/${tag}_${f}_bench.json | cut -c1-60) ... $(tail -1 $out/${tag}_${f}_bench.json | grep -o '"ms_per_step": [0-9.]*')"; done
