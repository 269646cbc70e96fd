"""Which large GEMM launches of a bench configuration run on the run-time (generic) epilogue?  UIA_TRACE_GENERIC=1 python tools/trace_generic_masks.py [bench args]"""
import os, sys, runpy
os.environ["UIA_TRACE_GENERIC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--also-streams", "0"] + sys.argv[1:]
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
from uia_hip import ops
names = {1: "BIAS", 2: "AUX_OUT", 4: "GELU", 8: "DGELU", 16: "RESID", 32: "RESIDT", 64: "OUT32", 128: "OUTT", 256: "RESID_LN", 512: "ROWSUM", 1024: "LNFOLD", 2048: "QUICK", 4096: "RESID_LO", 8192: "OUT_LO"}
for (mask, M, N, K, cfg), n in sorted(ops._TRACE_GENERIC.items(), key=lambda kv: -kv[1]):
    bits = "|".join(v for k, v in names.items() if mask >= 0 and mask & k) if mask >= 0 else "run-time features (alpha / row remap / dropout)"
    print(f"{n:4d} launches  mask {mask:6d} = {bits}   M={M} N={N} K={K} cfg {cfg}")
