"""bench.py against a measurement build of the library: python scratch/bench_ablation.py "-DFRCNN_H2_PP_MIN_K=512" -- <bench.py arguments>"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE)]
i = sys.argv.index("--")
defs, rest = sys.argv[1:i], sys.argv[i + 1:]
import ablation_lib
ablation_lib.use(extra=defs)
sys.argv = ["bench.py"] + rest
import bench
bench.main()
