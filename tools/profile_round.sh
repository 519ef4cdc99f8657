#!/bin/bash
# Produce the per-round profile artefacts on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02a
# writes gpurun_out/<tag>_{bench.json (the compact stdout line),bench_detail.json (the full record),bench_under_rocprof.json,
#   kernel_stats.csv,pmc_summary.txt,spmv_hbm_traffic.json}
# Every run includes the fp64-record repeat (roofline_general), so the kernel stats and the PMC passes cover
# spmv_pair_kernel (headline operator) AND spmv_sell_kernel (the format any mesh gets).
set -u
TAG=${1:-round}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
( time python3 bench.py --detail-path "$OUT/${TAG}_bench_detail.json" 2>/dev/null | tail -1 > "$OUT/${TAG}_bench.json" ) 2> "$OUT/${TAG}_bench_wall.txt"
rm -rf /tmp/prof_stats /tmp/prof_fetch /tmp/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 bench.py --cpu-iters 0 --skip-blas1 --traffic off --skip-permuted --skip-unstructured --skip-unstructured3d --skip-configs --detail-path "$OUT/${TAG}_bench_under_rocprof_detail.json" 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_under_rocprof.json"
cp "$(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv"
# the stand-alone SpMV block in a process of its own: the AverageNs of spmv_canon_tile_kernel<false,...> and
# spmv_sell_kernel<true, false,...> in this file is what `all_launches_mean_ms` of the JSON beside it must agree with
rm -rf /tmp/prof_spmv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_spmv -- python3 bench.py --spmv-only 2>/dev/null | tail -1 > "$OUT/${TAG}_spmv_only_under_rocprof.json"
cp "$(find /tmp/prof_spmv -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_spmv_only_kernel_stats.csv"
# ... and the same for the tetrahedral mesh (the fp64-record kernel again: a run of its own keeps its average apart)
rm -rf /tmp/prof_tets
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tets -- python3 bench.py --spmv-only --spmv-what tets 2>/dev/null | tail -1 > "$OUT/${TAG}_spmv_only_tets_under_rocprof.json"
cp "$(find /tmp/prof_tets -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_spmv_only_tets_kernel_stats.csv"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_fetch -- python3 bench.py --cpu-iters 0 --skip-blas1 --traffic off --skip-permuted --skip-unstructured --skip-unstructured3d --skip-configs --steps 20 --warmup 2 --spinup-seconds 0 --min-seconds 0 --detail-path '' > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_write -- python3 bench.py --cpu-iters 0 --skip-blas1 --traffic off --skip-permuted --skip-unstructured --skip-unstructured3d --skip-configs --steps 20 --warmup 2 --spinup-seconds 0 --min-seconds 0 --detail-path '' > /dev/null 2>&1
FMT=$(python3 -c "import json;print(json.load(open('$OUT/${TAG}_bench_detail.json'))['roofline']['record_format'])")
ALG=$(python3 -c "import json;print(json.load(open('$OUT/${TAG}_bench_detail.json'))['roofline']['algorithmic_bytes_8d'])")
FBY=$(python3 -c "import json;print(json.load(open('$OUT/${TAG}_bench_detail.json'))['roofline']['bytes_per_launch'])")
python3 tools/pmc_summary.py /tmp/prof_fetch /tmp/prof_write --traffic-json "$OUT/${TAG}_spmv_hbm_traffic.json" \
    --record-format "$FMT" --algorithmic-bytes "$ALG" --format-bytes "$FBY" > "$OUT/${TAG}_pmc_summary.txt"
python3 -c "
import json
line = open('$OUT/${TAG}_bench.json').read().strip()
print('stdout line:', len(line), 'characters;', 'strict JSON' if json.loads(line) else '')
d = json.load(open('$OUT/${TAG}_bench_detail.json'))
print('phase seconds', d.get('phase_seconds'))
print('CG it/s', d['value'], 'ms/step', d['ms_per_step'], d['timing'])
print('roofline', {k: d['roofline'][k] for k in ('kernel', 'achieved', 'frac', 'traffic', 'avg_launch_ms', 'frac_8d', 'record_format')})
print('spmv', {k: {m: (v[m]['median_ms'], round(v[m]['frac_8d'], 3), round(v[m]['frac_streamed'], 3)) for m in ('back_to_back', 'rotating_3_pairs')} for k, v in (d.get('spmv') or {}).items() if isinstance(v, dict) and 'back_to_back' in v})
u = d.get('roofline_unstructured3d') or {}
print('unstructured3d', {k: u.get(k) for k in ('frac', 'traffic_over_8d_bytes', 'ell_padding_ratio', 'tail_nnz', 'cg_iter_per_s', 'residual_after_20_iterations_rel_diff_vs_file_order')}, (u.get('spmv') or {}).get('rotating_3_pairs', {}).get('frac_8d'))
print('roofline_general', d.get('roofline_general'))
print('general', d.get('general_mesh_path'))
print('cpu', d['cpu_baseline'] and d['cpu_baseline'].get('value'))
"
head -12 "$OUT/${TAG}_kernel_stats.csv" | cut -c1-160
cat "$OUT/${TAG}_spmv_hbm_traffic.json"
