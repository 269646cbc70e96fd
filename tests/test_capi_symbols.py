"""CPU: libuia_hip.so loads and exports exactly the entry points include/uia_hip.h declares, and the ctypes prototype table
covers all of them (no compute calls: there is no GPU here)."""
import ctypes
import os
import sys
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "uia_hip.h")


def declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uia_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_functions():
    names = declared()
    assert len(names) >= 29 and "uia_gemm" in names and "uia_mona_spatial_bwd" in names and "uia_allreduce_sum" in names


def test_library_exports_every_declared_symbol():
    from uia_hip import _lib
    assert os.path.exists(_lib.LIB_PATH), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared() if not hasattr(handle, n)]
    assert not missing, f"declared in include/uia_hip.h but not exported: {missing}"


def test_ctypes_table_matches_header():
    from uia_hip import _lib
    names = set(declared())
    table = set(_lib.PROTOTYPES)
    assert table == names, f"only in header: {sorted(names - table)}; only in ctypes table: {sorted(table - names)}"


def test_descriptor_struct_sizes_match_c_layout():
    """The ctypes mirrors of uia_gemm_desc / uia_attn_desc / uia_mona_spatial_desc must have the C layout (LP64)."""
    import subprocess
    import tempfile
    from uia_hip import _lib
    assert ctypes.sizeof(_lib.GemmDesc) == 400 and ctypes.sizeof(_lib.PackDesc) == 64 and _lib.GemmDesc.a_drop_out.offset == 296
    assert ctypes.sizeof(_lib.AttnDesc) == 160
    assert ctypes.sizeof(_lib.MonaSpatialDesc) % 8 == 0 and ctypes.sizeof(_lib.MonaSpatialDesc) == 296
    assert ctypes.sizeof(_lib.MonaFusedDesc) == 432 and _lib.MonaFusedDesc.D.offset == 296
    # ... and the same numbers from the C compiler itself (sizes and the offset of the last field of the GEMM descriptor)
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "s.c"), os.path.join(td, "s")
        open(src, "w").write('#include <stdio.h>\n#include <stddef.h>\n#include "uia_hip.h"\nint main(void){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(uia_gemm_desc), '
                             'sizeof(uia_attn_desc), sizeof(uia_mona_spatial_desc), offsetof(uia_gemm_desc, ln_flag_limit), sizeof(uia_mona_fused_desc), '
                             'offsetof(uia_mona_fused_desc, t_out), sizeof(uia_lora_rank_desc), offsetof(uia_lora_rank_desc, seed), sizeof(uia_wgrad_group_desc), '
                             'offsetof(uia_wgrad_group_desc, drop_seed), offsetof(uia_wgrad_group_desc, drop_col0)); return 0; }\n')
        subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", exe, src], check=True)
        got = [int(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    assert got == [ctypes.sizeof(_lib.GemmDesc), ctypes.sizeof(_lib.AttnDesc), ctypes.sizeof(_lib.MonaSpatialDesc), _lib.GemmDesc.ln_flag_limit.offset,
                   ctypes.sizeof(_lib.MonaFusedDesc), _lib.MonaFusedDesc.t_out.offset, ctypes.sizeof(_lib.LoraRankDesc), _lib.LoraRankDesc.seed.offset,
                   ctypes.sizeof(_lib.WgradGroupDesc), _lib.WgradGroupDesc.drop_seed.offset, _lib.WgradGroupDesc.drop_col0.offset]


def test_integration_md_ctypes_stub_is_the_real_descriptor():
    """INTEGRATION.md quotes a ctypes mirror of uia_gemm_desc for outside users: its field list must be _lib.GemmDesc's, name for name and
    type for type (round 3's copy was 80 bytes short: a maintainer binding with it would have handed uia_gemm a struct whose tail — ln_flag,
    a_drop_out, splitk_ws, A2 — is whatever follows it in memory)."""
    import re
    from uia_hip import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    body = text[text.index("class GemmDesc(C.Structure):"):text.index("lib.uia_gemm.restype")]
    body = "\n".join(line.split("#")[0] for line in body.splitlines())
    quoted = re.findall(r'\("(\w+)",\s*C\.(c_\w+)\)', body)
    quoted = [(n, getattr(ctypes, t)) for n, t in quoted]           # c_int64 and c_long are one type on LP64: compare the types, not their names
    real = list(_lib.GemmDesc._fields_)
    assert quoted == real, f"INTEGRATION.md GemmDesc differs from _lib.GemmDesc: {[(a, b) for a, b in zip(quoted, real) if a != b][:3]} (lengths {len(quoted)} / {len(real)})"
    m = re.search(r"sizeof == (\d+)", text)
    assert m and int(m.group(1)) == ctypes.sizeof(_lib.GemmDesc)


def test_error_path_without_gpu():
    from uia_hip import _lib
    lib = _lib.lib()
    assert lib.uia_version() >= 100
    assert lib.uia_gemm(None, 1, None, 0) != 0                       # null descriptor is rejected before any launch
    assert b"null descriptor" in lib.uia_last_error()


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header must compile as C (gcc), with no C++ or torch types."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "uia_hip.h"\nint main(void){ return (int)sizeof(uia_gemm_desc) == 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", str(src)], check=True)


@pytest.mark.timeout(900)
def test_c_abi_error_paths_under_host_asan_ubsan():
    """SURVEY §5 "sanitizers": the whole library built with HOST AddressSanitizer + UBSan (`make asan`; device code is not instrumented
    — GPU ASan needs XNACK) and every argument-validation / error path of the C ABI driven on this CPU-only box (tests/asan_driver.py):
    null descriptors, empty / misaligned / undersized operands, unknown tile configs, communicator misuse, the no-device launch failure.
    A sanitizer report aborts the child process."""
    import glob
    import subprocess
    csrc = os.path.join(ROOT, "nextgen-uia_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "-j8", "asan"], check=True, capture_output=True)
    lib = os.path.join(ROOT, "nextgen-uia_amd", "uia_hip", "libuia_hip_asan.so")
    rts = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    assert os.path.exists(lib) and rts, "sanitizer build or runtime missing"
    env = dict(os.environ, LD_PRELOAD=rts[-1], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")                      # error paths only: never touch a GPU even if one is present
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_driver.py"), lib], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "no sanitizer report" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
