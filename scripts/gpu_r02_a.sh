#!/bin/bash
# Round-2 first GPU pass: full parity suite (with the printed error figures), default bench, chain / ddpm modes.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r02_a}
python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^\[|passed|failed|rror" gpurun_out/${TAG}_pytest.log | tail -60
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/${TAG}_bench.err
python scripts/bench_brief.py gpurun_out/${TAG}_bench.json
python bench.py --mode chain --steps 2 --warmup 1 > gpurun_out/${TAG}_bench_chain.json 2> gpurun_out/${TAG}_bench_chain.err; echo "chain rc=$?"; tail -3 gpurun_out/${TAG}_bench_chain.err
python bench.py --mode ddpm --batch 313 --steps 1 --warmup 0 > gpurun_out/${TAG}_bench_ddpm313.json 2> gpurun_out/${TAG}_bench_ddpm313.err; echo "ddpm rc=$?"; tail -3 gpurun_out/${TAG}_bench_ddpm313.err
python - <<PY
import json
for f in ("${TAG}_bench", "${TAG}_bench_chain", "${TAG}_bench_ddpm313"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, round(d["value"], 1), d["unit"], round(d["ms_per_step"], 1), "ms/step", {k: d[k] for k in ("end_to_end_mfma_frac", "p50_single_clip_latency_ms", "p50_chained_window_latency_ms") if k in d})
        if "chain_window_latency" in d: print(json.dumps(d["chain_window_latency"])[:600])
        if "cpu_baseline" in d: print(d["cpu_baseline"]["value"], d["cpu_baseline_config1"]["value"], d["cpu_baseline_config1"]["sample"][-40:])
    except Exception as e: print(f, "ERR", e)
PY
