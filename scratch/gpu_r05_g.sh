# round 5, call G: full GPU suite + default bench line at the state "light tile boundary + trunk as planes by default"
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_g}
mkdir -p gpurun_out
rm -f gpurun_out/fullsize_parity.txt
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log
cp gpurun_out/fullsize_parity.txt gpurun_out/${TAG}_fullsize_parity.txt
( time timeout 600 python bench.py --steps 20 --warmup 5 --layer-report gpurun_out/${TAG}_layer_table.txt 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json ) 2> gpurun_out/${TAG}_bench_wall.txt
tail -3 gpurun_out/${TAG}_pytest.log; cat gpurun_out/${TAG}_bench_wall.txt | tail -4; python -c "
import json; d=json.loads(open('gpurun_out/${TAG}_bench.json').read()); r=d.get('roofline') or {}; print(d['value'], d['ms_per_step'], (d.get('x3_variant') or {}).get('value'), (d.get('f32_mfma_variant') or {}).get('value'), r.get('frac'), r.get('mfma_frac'), r.get('sclk_mhz'), r.get('socket_w'), d.get('latency_ms_batch1')); print({k: (v.get('value'), v.get('ms_per_step'), v.get('roofline_frac')) for k, v in (d.get('other_configs') or {}).items()})"
