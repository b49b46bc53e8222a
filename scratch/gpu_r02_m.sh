# x3: unit tests, network / full-size parity, bench (with the f32 variant timed in the same run)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_dense_gpu.py -m gpu -q -x -k "x3" -s 2>&1 | grep -E "^x3|passed|failed|Error" ) > gpurun_out/m_x3_unit.txt
(timeout 900 python -m pytest tests/test_network_gpu.py tests/test_fullsize_gpu.py tests/test_boundary_gpu.py -m gpu -q -s 2>&1 | grep -E "^c[23] |passed|failed|Error|assert" | cut -c1-260) > gpurun_out/m_parity.txt
timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/m_bench.json
cat gpurun_out/m_x3_unit.txt; cat gpurun_out/m_parity.txt; python -c "
import json; d=json.loads(open('gpurun_out/m_bench.json').read()); print(d['value'], d['ms_per_step'], d.get('f32_mfma_variant'), d['roofline']['achieved'], d['roofline']['frac'])"
