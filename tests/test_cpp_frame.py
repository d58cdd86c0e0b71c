"""Builds and runs the C++ host-side mirror's tests (tests/cpp/*.cpp over include/rdf_frame.hpp).

test_plan: plan builders (src/operation/scalar.rs) — CPU only.
test_frame: the reference's #[test]s for the hot path restated in C++ and run on the device, plus the fused
            batch loop against the unfused oracle — GPU."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rust_dataframe_amd")
ORA = os.path.join(ROOT, "oracle")


def build(name, with_oracle):
    out = os.path.join(tempfile.gettempdir(), f"rdf_{name}_{os.getpid()}")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", ORA,
           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", out, "-L", PKG, "-lrdf_mi355x", f"-Wl,-rpath,{PKG}"]
    if with_oracle:
        if not os.path.exists(os.path.join(ORA, "librdf_oracle.so")):
            subprocess.check_call(["make", "-C", ORA, "-s"])
        cmd += ["-L", ORA, "-lrdf_oracle", f"-Wl,-rpath,{ORA}"]
    subprocess.check_call(cmd)
    return out


def run(exe, *args):
    p = subprocess.run([exe, *args], cwd=ROOT, capture_output=True, text=True, timeout=600)
    print(p.stdout[-4000:], p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-4000:]
    assert " 0 failed" in p.stdout


def test_plan_builders_cpp():
    run(build("test_plan", False))


def test_jit_signatures_cpp():
    """rdf_jit.cpp's host logic (signature -> template argument, refusals, code-object symbol lookup) — built with hipcc because the
    file includes the HIP runtime's headers; nothing in it touches a device."""
    out = os.path.join(tempfile.gettempdir(), f"rdf_test_jit_sig_{os.getpid()}")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    subprocess.check_call([hipcc, "-x", "hip", "--offload-arch=gfx950", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "rust_dataframe_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "test_jit_sig.cpp"), "-o", out, "-ldl"])
    run(out)


def test_stage_copy_cpp():
    """packed_copy (csrc/rdf_stage_copy.h): host chunks into / out of the staging buffer, cut by bytes over threads — every staged byte
    exactly once, nothing else touched, for a handful of huge pieces, thousands of tiny batches, empty and unstaged items."""
    out = os.path.join(tempfile.gettempdir(), f"rdf_test_stage_copy_{os.getpid()}")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_stage_copy.cpp"), "-o", out])
    run(out)


def test_sort_bucket_map_cpp():
    """The bucket map of Float64 sort keys (csrc/rdf_sort_map.h: the function the kernels run, under the planner the host runs) on
    the CPU: monotone over sorted columns of every shape and across every seam, inside its buckets, the fullest bucket inside the
    LDS finish; columns no map can split are refused by the plan."""
    out = os.path.join(tempfile.gettempdir(), f"rdf_test_sort_map_{os.getpid()}")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", os.path.join(ROOT, "tests", "cpp", "test_sort_map.cpp"), "-o", out])
    run(out)


def test_join_place_cpp():
    """The scan that lays out the equi-join's table of distinct build keys (csrc/rdf_join_place.h, the code the kernels run) on the CPU:
    run summaries combine associatively, tiles cut any way place every key where sequential linear probing would, a probe walking from a
    key's home slot meets no empty slot before the key."""
    out = os.path.join(tempfile.gettempdir(), f"rdf_test_join_place_{os.getpid()}")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", os.path.join(ROOT, "tests", "cpp", "test_join_place.cpp"), "-o", out])
    run(out)


@pytest.mark.gpu
def test_frame_mirror_cpp():
    run(build("test_frame", True), os.path.join(ROOT, "tests", "golden", "uk_cities_with_headers.csv"),
        os.path.join(ROOT, "tests", "golden", "mixed_batches.arrow"))
