// clo_hip_jit_opts.h — the compiler options of a run-time compiled module (hiprtc): the fixed ones plus the
// caller's `compiler_opts`, which upstream hands to the OpenCL JIT with the kernel source
// (sort/clo_sort_abstract.c:173-179: ccl_program_build(prg, compiler_opts, ...)), so that a caller can define a
// macro there and use it inside `compare` / `get_key`. Host code only.
#ifndef CLO_HIP_JIT_OPTS_H
#define CLO_HIP_JIT_OPTS_H

#include <string>
#include <vector>

namespace {

// Split on white space. "-D NAME", "-U NAME", "-I DIR" (the separated spelling OpenCL allows) are joined; OpenCL's
// own switches (-cl-..., which mean nothing to a HIP compiler) are dropped; everything else goes to the compiler as
// it stands — what it does not know it refuses, and its log says so.
inline std::vector<std::string> clo_jit_options(const char* compiler_opts) {
	std::vector<std::string> out = { "--offload-arch=gfx950", "-O3", "-std=c++17" };
	const std::string s = compiler_opts ? compiler_opts : "";
	size_t i = 0;
	std::string pending;
	while (i < s.size()) {
		while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n')) ++i;
		size_t j = i;
		while (j < s.size() && s[j] != ' ' && s[j] != '\t' && s[j] != '\n') ++j;
		if (j == i) break;
		std::string tok = s.substr(i, j - i);
		i = j;
		if (!pending.empty()) { out.push_back(pending + tok); pending.clear(); continue; }
		if (tok == "-D" || tok == "-U" || tok == "-I") { pending = tok; continue; }
		if (tok.compare(0, 4, "-cl-") == 0) continue;
		out.push_back(tok);
	}
	if (!pending.empty()) out.push_back(pending);   // ("-D" with nothing behind it: the compiler says what is wrong)
	return out;
}

}  // namespace

#endif
