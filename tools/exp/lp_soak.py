"""LP-level reproducibility soak: tests/test_determinism_gpu.py::test_lp_pass_is_reproducible_run_to_run_with_every_overlap at arbitrary shapes and pass counts.
GPU box: BFSR_LP_PASSES=120 python tools/exp/lp_soak.py 8,64,96 8,32,96 4,16,160"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_determinism_gpu as t
for a in sys.argv[1:]:
    shape = tuple(int(v) for v in a.split(","))
    t0 = time.time()
    t.test_lp_pass_is_reproducible_run_to_run_with_every_overlap(*shape)
    print("LP soak (scale, batch, lr) = %s: %s passes bit-identical (%.0f s)" % (shape, os.environ.get("BFSR_LP_PASSES", "4"), time.time() - t0), flush=True)
