// rdf_jit.cpp — specialised kernels for program shapes the ahead-of-time catalog does not hold.
//
// The catalog (rdf_spec.hip, rdf_spec_shapes.hip) instantiates spec_kernel<Prog<..>> for the shapes the reference's plan
// builders produce: 1329 programs.  Anything else — a four-level tree, a cast below the leaves, a unary function inside an
// arithmetic chain — used to fall to the general evaluator (rdf_eval.hip), which interprets byte code in generic 64-bit
// arithmetic and is bound by instruction issue at 0.05-0.4 of the HBM roofline where the specialised kernels reach 0.6-0.8.
// Here the SAME hand-written kernel template is instantiated for such a program at run time: the program's canonical signature
// (the string the catalog is keyed by) is turned back into the Prog<..> type it spells, `hipcc --cuda-device-only` compiles
// `template __global__ void rdfk::spec_kernel<P>(SpecArgs)` from the library's own kernel sources with build()'s flags for the
// device's architecture (about 1.5 s), and the code object stays loaded for the life of the process.  No kernel is generated:
// the device code is rdf_spec_kernel.hip.h / rdf_expr.hip.h as compiled by build(), only the template argument is new.
// The compiler runs as a child process: hiprtc in-process was tried first and crashed inside the compile for some programs
// whenever the host process had already loaded another ROCm release's libhiprtc / comgr under the same SONAME (PyTorch wheels
// bundle theirs) — a child process uses the toolchain the library was built with, whatever the host process holds.
// Whatever goes wrong (no hipcc, sources not next to the library, a program the template rejects) is remembered per signature
// and the call runs on the interpreter as before.  rdf_set_option("jit", 0) turns the path off.
#include <dlfcn.h>
#include <elf.h>
#include <fcntl.h>
#include <signal.h>
#include <hip/hip_runtime.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "rdf_device.h"

extern char** environ;

namespace rdfk {
namespace {

bool file_exists(const std::string& p) { FILE* f = std::fopen(p.c_str(), "rb"); if (f) std::fclose(f); return f != nullptr; }

// where the kernel sources and the compiler's headers are: RDF_JIT_SRC (the directory that holds rdf_spec_kernel.hip.h) or
// <directory of this library>/csrc; ROCM_PATH or /opt/rocm
struct Paths { std::string src, hipcc, stamp; bool ok = false; std::string why; };
const Paths& paths() {
    static Paths p;
    static bool tried = false;
    if (tried) return p;
    tried = true;
    if (const char* e = getenv("RDF_JIT_SRC")) p.src = e;
    else {
        Dl_info info;
        if (dladdr((const void*)&paths, &info) && info.dli_fname) {
            std::string lib = info.dli_fname;
            const size_t slash = lib.rfind('/');
            p.src = (slash == std::string::npos ? std::string(".") : lib.substr(0, slash)) + "/csrc";
            struct stat st;
            if (stat(lib.c_str(), &st) == 0) p.stamp = std::to_string((long long)st.st_size) + "." + std::to_string((long long)st.st_mtime);   // a rebuilt library invalidates cached code objects
        }
    }
    if (p.src.empty() || !file_exists(p.src + "/rdf_spec_kernel.hip.h")) { p.why = "kernel sources not found (" + p.src + "/rdf_spec_kernel.hip.h; set RDF_JIT_SRC)"; return p; }
    if (!file_exists(p.src + "/../../include/rdf_mi355x.h")) { p.why = "include/rdf_mi355x.h not found next to the kernel sources"; return p; }
    const std::string rocm = getenv("ROCM_PATH") ? getenv("ROCM_PATH") : "/opt/rocm";
    p.hipcc = getenv("RDF_JIT_HIPCC") ? getenv("RDF_JIT_HIPCC") : rocm + "/bin/hipcc";
    if (!file_exists(p.hipcc)) { p.why = p.hipcc + " not found (set RDF_JIT_HIPCC)"; return p; }
    p.ok = true;
    return p;
}

// ---- signature -> the type it spells (the grammar of Prog::sig() and the expression templates' sig(), rdf_expr.hip.h)
const char* tag_dtype(char t) {
    switch (t) {
        case 'd': return "RDF_F64"; case 'l': return "RDF_I64"; case 'u': return "RDF_U64"; case 'f': return "RDF_F32";
        case 'i': return "RDF_I32"; case 'j': return "RDF_U32"; case 'b': return "RDF_BOOL"; case 'a': return "RDF_I8";
        case 'h': return "RDF_U8"; case 's': return "RDF_I16"; case 't': return "RDF_U16"; default: return nullptr;
    }
}
bool number(const char*& p, std::string& out) {
    const char* b = p;
    while (*p >= '0' && *p <= '9') ++p;
    if (p == b || p - b > 4) return false;
    out.assign(b, p);
    return true;
}
bool expr(const char*& p, std::string& out, int depth = 0) {
    if (depth > 32) return false;
    if (*p == '-') { ++p; out = "rdfk::None"; return true; }
    if (*p == 'c' || *p == 'k') {
        const bool col = *p == 'c';
        if (!(p[1] >= '0' && p[1] <= '9')) return false;
        const char* dt = tag_dtype(p[2]);
        if (!dt) return false;
        out = std::string(col ? "rdfk::Col<" : "rdfk::Imm<") + p[1] + ", " + dt + ">";
        p += 3;
        return true;
    }
    const char open = *p;
    if (open != '(' && open != '[' && open != '{') return false;
    ++p;
    std::string n, a, b;
    if (!number(p, n) || *p != ' ') return false;     // (a letter here is a runtime-operator slot of a shape kernel: not this grammar)
    ++p;
    if (!expr(p, a, depth + 1)) return false;
    if (open == '(') {
        if (*p != ' ') return false;
        ++p;
        if (!expr(p, b, depth + 1) || *p != ')') return false;
        ++p;
        out = "rdfk::Bin<" + n + ", " + a + ", " + b + ">";
    } else if (open == '[') {
        if (*p != ']') return false;
        ++p;
        out = "rdfk::Un<" + n + ", " + a + ">";
    } else {
        if (*p != '}') return false;
        ++p;
        out = "rdfk::Cast<" + n + ", " + a + ">";
    }
    return true;
}
// the grouped sink's programs (rdf_gspec_kernel.hip.h): "G<g>;P:<pred>;K:<group id>;V:<v0>;<v1>;...;"
bool gprog_type(const char* sig, std::string& out) {
    const char* p = sig;
    std::string g, pred, key, vals, v;
    if (*p != 'G') return false;
    ++p;
    if (!number(p, g) || strncmp(p, ";P:", 3) != 0) return false;
    p += 3;
    if (!expr(p, pred) || strncmp(p, ";K:", 3) != 0) return false;
    p += 3;
    if (!expr(p, key) || strncmp(p, ";V:", 3) != 0) return false;
    p += 3;
    int nv = 0;
    while (*p) {
        if (!expr(p, v) || *p != ';') return false;
        ++p;
        vals += ", " + v;
        ++nv;
    }
    if (nv < 1) return false;
    out = "rdfk::GProg<" + g + ", " + pred + ", " + key + vals + ">";
    return true;
}
bool prog_type(const char* sig, std::string& out) {   // "P:<pred>;V:<v0>;<v1>;S:<sink>"
    const char* p = sig;
    std::string pred, v0, v1, sink;
    if (strncmp(p, "P:", 2) != 0) return false;
    p += 2;
    if (!expr(p, pred) || strncmp(p, ";V:", 3) != 0) return false;
    p += 3;
    if (!expr(p, v0) || *p != ';') return false;
    ++p;
    if (!expr(p, v1) || strncmp(p, ";S:", 3) != 0) return false;
    p += 3;
    if (!number(p, sink) || *p) return false;
    out = "rdfk::Prog<" + pred + ", " + v0 + ", " + v1 + ", " + sink + ">";
    return true;
}

struct Entry { JitKernel k{nullptr, 0, 0}; bool failed = false; };
std::mutex g_mu;
std::map<std::string, Entry> g_kernels;
int g_compiled = 0;

bool read_file(const std::string& path, std::vector<char>& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n > 0 && std::fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    std::fclose(f);
    return ok;
}
// the mangled name of the one spec_kernel instantiation in a code object (ELF64, little endian): its dynamic symbol table
bool kernel_symbol(const std::vector<char>& elf, const char* prefix, std::string& name) {
    if (elf.size() < sizeof(Elf64_Ehdr) || memcmp(elf.data(), ELFMAG, SELFMAG) != 0) return false;
    const Elf64_Ehdr* eh = (const Elf64_Ehdr*)elf.data();
    if (eh->e_shoff == 0 || eh->e_shoff + (uint64_t)eh->e_shnum * sizeof(Elf64_Shdr) > elf.size()) return false;
    const Elf64_Shdr* sh = (const Elf64_Shdr*)(elf.data() + eh->e_shoff);
    for (int i = 0; i < eh->e_shnum; ++i) {
        if (sh[i].sh_type != SHT_DYNSYM && sh[i].sh_type != SHT_SYMTAB) continue;
        if (sh[i].sh_link >= eh->e_shnum) continue;
        const Elf64_Shdr& str = sh[sh[i].sh_link];
        if (sh[i].sh_offset + sh[i].sh_size > elf.size() || str.sh_offset + str.sh_size > elf.size()) continue;
        const Elf64_Sym* sym = (const Elf64_Sym*)(elf.data() + sh[i].sh_offset);
        const size_t nsym = sh[i].sh_size / sizeof(Elf64_Sym);
        for (size_t k = 0; k < nsym; ++k) {
            if (ELF64_ST_TYPE(sym[k].st_info) != STT_FUNC || sym[k].st_name >= str.sh_size) continue;
            const char* s = elf.data() + str.sh_offset + sym[k].st_name;
            if (strncmp(s, prefix, strlen(prefix)) == 0) { name = s; return true; }
        }
    }
    return false;
}

// RDF_JIT_CACHE=<directory>: code objects are kept there across processes, keyed by signature, architecture and the library's
// size + modification time (a new process then loads a known shape in milliseconds instead of compiling it again)
std::string cache_file(const char* sig, const std::string& arch, const Paths& ps) {
    const char* dir = getenv("RDF_JIT_CACHE");
    if (!dir || !*dir) return std::string();
    uint64_t h = 1469598103934665603ull;
    for (const std::string& part : {std::string(sig), arch, ps.stamp})
        for (unsigned char c : part + "|") { h ^= c; h *= 1099511628211ull; }
    char name[32];
    std::snprintf(name, sizeof name, "%016llx.hsaco", (unsigned long long)h);
    (void)mkdir(dir, 0700);
    return std::string(dir) + "/" + name;
}

// one run of the compiler: the kernel source in a scratch directory, hipcc as a child process with build()'s flags (csrc/Makefile),
// device side only, the code object itself (no offload bundle)
bool compile(const Paths& ps, const std::string& arch_opt, const std::string& type, bool grouped, std::vector<char>& code, std::string& why) {
    const char* tmp = getenv("TMPDIR");
    std::string dir_s = std::string(tmp && *tmp ? tmp : "/tmp") + "/rdf_jit_XXXXXX";
    if (!mkdtemp(&dir_s[0])) { why = "mkdtemp failed under " + dir_s; return false; }
    const std::string dir = dir_s, src_path = dir + "/k.hip", obj_path = dir + "/k.hsaco", log_path = dir + "/k.log";
    bool ok = false;
    do {
        FILE* f = std::fopen(src_path.c_str(), "wb");
        if (!f) { why = "cannot write " + src_path; break; }
        if (grouped)
            std::fprintf(f, "#include \"rdf_gspec_kernel.hip.h\"\nusing P = %s;\ntemplate __global__ void rdfk::gspec_kernel<P>(const rdfk::GSpecArgs);\n"
                            "extern \"C\" __global__ void rdf_jit_meta(int* out) { out[0] = P::R; out[1] = P::NV; out[2] = P::G; out[3] = P::NC; }\n", type.c_str());
        else
            std::fprintf(f, "#include \"rdf_spec_kernel.hip.h\"\nusing P = %s;\ntemplate __global__ void rdfk::spec_kernel<P>(const rdfk::SpecArgs);\n"
                            "extern \"C\" __global__ void rdf_jit_meta(int* out) { out[0] = P::R; out[1] = P::U; out[2] = P::W; out[3] = P::NC; }\n", type.c_str());
        std::fclose(f);
        const std::string inc = "-I" + ps.src;
        std::vector<std::string> argv_s = {ps.hipcc, arch_opt, "--cuda-device-only", "--no-gpu-bundle-output", "-O3", "-std=c++17", "-ffp-contract=off", inc, "-c", src_path, "-o", obj_path};
        std::vector<char*> argv;
        for (std::string& a : argv_s) argv.push_back(&a[0]);
        argv.push_back(nullptr);
        posix_spawn_file_actions_t fa;
        posix_spawn_file_actions_init(&fa);
        posix_spawn_file_actions_addopen(&fa, 1, log_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
        posix_spawn_file_actions_adddup2(&fa, 1, 2);
        pid_t pid = 0;
        const int rc = posix_spawn(&pid, ps.hipcc.c_str(), &fa, nullptr, argv.data(), environ);
        posix_spawn_file_actions_destroy(&fa);
        if (rc != 0) { why = "cannot start " + ps.hipcc; break; }
        // the caller's thread waits for the compiler, but not for ever: RDF_JIT_TIMEOUT seconds (default 120), then the child is killed
        int status = 0;
        {
            const char* te = getenv("RDF_JIT_TIMEOUT");
            const long limit_ms = (te && atol(te) > 0 ? atol(te) : 120) * 1000L;
            long waited_ms = 0;
            bool done = false;
            while (!done) {
                const pid_t r = waitpid(pid, &status, WNOHANG);
                if (r == pid) done = true;
                else if (r < 0 && errno != EINTR) { status = -1; done = true; }
                else if (waited_ms >= limit_ms) {
                    (void)kill(pid, SIGKILL);
                    while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {}
                    why = "the compiler did not finish in time";
                    status = -1;
                    done = true;
                } else { usleep(5000); waited_ms += 5; }
            }
        }
        if (status == -1 && !why.empty()) break;
        if (!WIFEXITED(status) || WEXITSTATUS(status) != 0 || !read_file(obj_path, code)) {
            std::vector<char> log;
            (void)read_file(log_path, log);
            why = "compilation failed: " + std::string(log.begin(), log.begin() + std::min<size_t>(log.size(), 1500));
            break;
        }
        ok = true;
    } while (false);
    if (!getenv("RDF_JIT_KEEP")) { (void)unlink(src_path.c_str()); (void)unlink(obj_path.c_str()); (void)unlink(log_path.c_str()); (void)rmdir(dir.c_str()); }
    return ok;
}

bool build(const char* sig, Entry& e, std::string& why) {
    const Paths& ps = paths();
    if (!ps.ok) { why = ps.why; return false; }
    std::string type;
    const bool grouped = sig[0] == 'G';
    if (!(grouped ? gprog_type(sig, type) : prog_type(sig, type))) { why = "not an exact-program signature"; return false; }
    static const bool dbg = getenv("RDF_DEBUG_JIT") != nullptr;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { why = "no device"; return false; }
    const std::string arch_opt = std::string("--offload-arch=") + prop.gcnArchName;     // "gfx950:sramecc+:xnack-"
    const std::string cached = cache_file(sig, prop.gcnArchName, ps);
    std::vector<char> code;
    if (!cached.empty() && read_file(cached, code)) {
        if (dbg) fprintf(stderr, "[rdf] jit: %s from %s\n", sig, cached.c_str());
    } else {
        if (!compile(ps, arch_opt, type, grouped, code, why)) return false;
        if (!cached.empty()) {   // written under another name and renamed: a concurrent reader never sees half a file
            const std::string tmp = cached + "." + std::to_string((long)getpid());
            FILE* c = std::fopen(tmp.c_str(), "wb");
            if (c) {
                const bool w = std::fwrite(code.data(), 1, code.size(), c) == code.size();
                std::fclose(c);
                if (!w || rename(tmp.c_str(), cached.c_str()) != 0) (void)unlink(tmp.c_str());
            }
        }
    }
    std::string kname;
    if (!kernel_symbol(code, grouped ? "_ZN4rdfk12gspec_kernel" : "_ZN4rdfk11spec_kernel", kname)) { why = "kernel symbol not found in the code object"; return false; }
    if (dbg) fprintf(stderr, "[rdf] jit: %zu bytes of code, loading %s\n", code.size(), kname.c_str());
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr, meta = nullptr;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess) { why = "hipModuleLoadData failed"; return false; }
    bool ok = false;
    do {
        if (hipModuleGetFunction(&fn, mod, kname.c_str()) != hipSuccess || hipModuleGetFunction(&meta, mod, "rdf_jit_meta") != hipSuccess) { why = "kernel not found in the code object"; break; }
        int* d_meta = nullptr;
        int h_meta[4] = {0, 0, 0, 0};
        if (hipMalloc((void**)&d_meta, 16) != hipSuccess) { why = "hipMalloc failed"; break; }
        void* margs[] = {(void*)&d_meta};
        const bool launched = hipModuleLaunchKernel(meta, 1, 1, 1, 1, 1, 1, 0, nullptr, margs, nullptr) == hipSuccess &&
                              hipMemcpy(h_meta, d_meta, 16, hipMemcpyDeviceToHost) == hipSuccess;
        (void)hipFree(d_meta);
        if (!launched || h_meta[0] <= 0) { why = "the kernel's tile size could not be read"; break; }
        e.k.fn = (void*)fn;
        e.k.rows_per_tile = grouped ? kEvalTile : 64 * h_meta[0];
        e.k.nvalues = grouped ? h_meta[1] : 0;
        ok = true;
    } while (false);
    if (!ok) (void)hipModuleUnload(mod);
    return ok;
}

}  // namespace

const JitKernel* jit_find(const char* sig) {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_kernels.find(sig);
    return it == g_kernels.end() || it->second.failed ? nullptr : &it->second.k;
}

const JitKernel* jit_spec_kernel(const char* sig) {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_kernels.find(sig);
    if (it != g_kernels.end()) return it->second.failed ? nullptr : &it->second.k;
    Entry e;
    std::string why;
    e.failed = !build(sig, e, why);
    static const bool dbg = getenv("RDF_DEBUG") != nullptr || getenv("RDF_DEBUG_JIT") != nullptr;
    if (dbg) {
        if (e.failed) fprintf(stderr, "[rdf] jit: %s -> interpreter (%s)\n", sig, why.c_str());
        else fprintf(stderr, "[rdf] jit: compiled spec_kernel<%s>, %d rows per wave iteration\n", sig, e.k.rows_per_tile);
    }
    if (!e.failed) ++g_compiled;
    auto ins = g_kernels.emplace(sig, e);
    return e.failed ? nullptr : &ins.first->second.k;
}

int jit_compiled_count() { std::lock_guard<std::mutex> lock(g_mu); return g_compiled; }

hipError_t jit_launch(const JitKernel& k, const SpecArgs& a, int grid, hipStream_t s) {
    SpecArgs copy = a;
    void* args[] = {(void*)&copy};
    return hipModuleLaunchKernel((hipFunction_t)k.fn, (unsigned)grid, 1, 1, kBlock, 1, 1, 0, s, args, nullptr);
}
hipError_t jit_launch_grouped(const JitKernel& k, const GSpecArgs& a, int grid, hipStream_t s) {
    GSpecArgs copy = a;
    void* args[] = {(void*)&copy};
    const unsigned lds = (unsigned)group_words(a.ngroups, k.nvalues) * 8u;     // the block's group table, as launch_gprog sizes it
    return hipModuleLaunchKernel((hipFunction_t)k.fn, (unsigned)grid, 1, 1, kBlock, 1, 1, lds, s, args, nullptr);
}

}  // namespace rdfk
