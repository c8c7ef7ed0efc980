"""How a caller's `compiler_opts` reach the run-time compiler (cl_ops_amd/csrc/hip/clo_hip_jit_opts.h, host code only):
split on white space, "-D NAME" / "-U NAME" / "-I DIR" joined, OpenCL's own -cl-* switches dropped, the rest passed as
it stands. Upstream hands the string to the OpenCL JIT with the source (sort/clo_sort_abstract.c:173-179). The header
is plain C++: compiled with g++ here, no GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MAIN = r'''
#include <cstdio>
#include "clo_hip_jit_opts.h"
int main(int argc, char** argv) {
	const std::vector<std::string> o = clo_jit_options(argc > 1 ? argv[1] : nullptr);
	for (const std::string& s : o) printf("[%s]\n", s.c_str());
	return 0;
}
'''


def _options(tmp_path, text):
    exe = tmp_path / "jit_opts"
    if not exe.exists():
        src = tmp_path / "main.cpp"
        src.write_text(MAIN)
        subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "cl_ops_amd", "csrc", "hip"), str(src), "-o", str(exe)])
    args = [str(exe)] + ([text] if text is not None else [])
    out = subprocess.run(args, capture_output=True, text=True, check=True).stdout
    return [l[1:-1] for l in out.splitlines()]


def test_compiler_opts_are_split_joined_and_filtered(tmp_path):
    fixed = ["--offload-arch=gfx950", "-O3", "-std=c++17"]
    assert _options(tmp_path, None) == fixed
    assert _options(tmp_path, "") == fixed
    assert _options(tmp_path, "  \t\n ") == fixed
    assert _options(tmp_path, "-DSHIFT=12") == fixed + ["-DSHIFT=12"]
    assert _options(tmp_path, "-D SHIFT=12  -U  OLD\t-I /some/dir") == fixed + ["-DSHIFT=12", "-UOLD", "-I/some/dir"]
    assert _options(tmp_path, "-cl-fast-relaxed-math -DA=1 -cl-mad-enable -w") == fixed + ["-DA=1", "-w"]
    assert _options(tmp_path, "-DA=1 -D") == fixed + ["-DA=1", "-D"]          # the compiler reports the dangling -D
    assert _options(tmp_path, "--not-an-option") == fixed + ["--not-an-option"]   # refused by hiprtc, with its log (GPU test)
