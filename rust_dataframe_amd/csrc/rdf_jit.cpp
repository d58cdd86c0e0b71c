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
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
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
    static Paths& p = *new Paths;          // (never destroyed: helper threads read it, see g_mu)
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

// What is known about a signature.  The CODE OBJECT belongs to the signature (compiled once per process, or read from the
// cache directory); a hipFunction_t belongs to the device it was loaded on, so the loaded kernels are keyed by (device, signature):
// a second thread on another device loads the same bytes there instead of borrowing the first device's handle.
enum CodeState { kCompiling = 0, kReady = 1, kFailed = 2 };
struct Code {
    int state = kCompiling;
    bool grouped = false;
    std::vector<char> bytes;
    std::string kname, why;
};
struct Loaded { JitKernel k{nullptr, 0, 0}; bool failed = false; };
// The shared state lives in heap objects that are never destroyed: helper threads (finish_code) may still hold it while the
// process runs its static destructors — a short-lived host that met a new shape and returned from main —, and objects with
// static storage duration would be torn down under them.  What ends the helpers is jit_shutdown() below.
std::mutex& g_mu = *new std::mutex;
std::condition_variable& g_cv = *new std::condition_variable;     // a signature left kCompiling
std::map<std::string, std::shared_ptr<Code>>& g_code = *new std::map<std::string, std::shared_ptr<Code>>;
std::map<std::pair<int, std::string>, Loaded>& g_loaded = *new std::map<std::pair<int, std::string>, Loaded>;
int g_compiled = 0, g_from_cache = 0, g_failed = 0, g_running = 0;
// helper threads are joinable and kept here (g_mu) together with a flag each sets as its last act; finished ones are reaped when
// the next one starts, the rest by jit_shutdown()
struct Helper { std::thread t; std::shared_ptr<std::atomic<bool>> done; };
std::vector<Helper>& g_helpers = *new std::vector<Helper>;
std::atomic<bool> g_shutdown{false};                              // set once: compilers in flight are killed, no new helper starts
constexpr int kMaxCompilers = 6;                                  // hipcc children at a time (a burst of new shapes queues behind them)

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

// ---- the code-object cache: ON by default, across processes
// Directory: RDF_JIT_CACHE (empty, "0" or "off": no cache), else $XDG_CACHE_HOME/rdf_mi355x/jit, else $HOME/.cache/rdf_mi355x/jit.
// A file is named by a hash of (signature, architecture, the library's size + mtime, the CONTENTS of the kernel headers the
// compiler reads, the compile flags) and carries its signature, which is compared on load: an edited header, another build or
// a hash collision can only miss.  A directory that is not the user's own, or that group / others may write, is not used —
// GPU code is loaded from it.
const char* kJitFlags = "-O3 -std=c++17 -ffp-contract=off --cuda-device-only";
uint64_t fnv(uint64_t h, const void* p, size_t n) { const unsigned char* c = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; } return h; }
uint64_t sources_hash(const Paths& ps) {
    static uint64_t h = 0;
    static bool done = false;
    if (done) return h;
    done = true;
    h = 1469598103934665603ull;
    for (const char* name : {"/rdf_spec_kernel.hip.h", "/rdf_gspec_kernel.hip.h", "/rdf_expr.hip.h", "/rdf_common.hip.h", "/rdf_device.h", "/rdf_sort_map.h", "/../../include/rdf_mi355x.h"}) {
        std::vector<char> buf;
        if (read_file(ps.src + name, buf)) h = fnv(h, buf.data(), buf.size());
        h = fnv(h, name, strlen(name));
    }
    return h;
}
struct CacheDir { std::string dir, why; };
const CacheDir& cache_dir() {
    static CacheDir& c = *new CacheDir;    // (never destroyed: helper threads read it)
    static bool tried = false;
    if (tried) return c;
    tried = true;
    const char* e = getenv("RDF_JIT_CACHE");
    std::string d;
    if (e) {
        if (!*e || strcmp(e, "0") == 0 || strcmp(e, "off") == 0) { c.why = "switched off (RDF_JIT_CACHE)"; return c; }
        d = e;
    } else {
        const char* x = getenv("XDG_CACHE_HOME");
        const char* h = getenv("HOME");
        if (x && *x) d = std::string(x) + "/rdf_mi355x/jit";
        else if (h && *h) d = std::string(h) + "/.cache/rdf_mi355x/jit";
        else { c.why = "no RDF_JIT_CACHE, XDG_CACHE_HOME or HOME"; return c; }
    }
    for (size_t i = 1; i <= d.size(); ++i)                                     // mkdir -p, every new level private
        if (i == d.size() || d[i] == '/') { const std::string part = d.substr(0, i); (void)mkdir(part.c_str(), 0700); }
    struct stat st;
    if (stat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) { c.why = "cannot create " + d; return c; }
    if (st.st_uid != geteuid()) { c.why = d + " belongs to another user"; return c; }
    if (st.st_mode & (S_IWGRP | S_IWOTH)) { c.why = d + " is writable by group / others"; return c; }
    c.dir = d;
    return c;
}
std::string cache_file(const char* sig, const std::string& arch, const Paths& ps) {
    const CacheDir& cd = cache_dir();
    if (cd.dir.empty()) return std::string();
    uint64_t h = 1469598103934665603ull;
    const uint64_t sh = sources_hash(ps);
    for (const std::string& part : {std::string(sig), arch, ps.stamp, std::string(kJitFlags)}) { h = fnv(h, part.data(), part.size()); h = fnv(h, "|", 1); }
    h = fnv(h, &sh, 8);
    char name[32];
    std::snprintf(name, sizeof name, "%016llx.hsaco", (unsigned long long)h);
    return cd.dir + "/" + name;
}
const char kCacheMagic[8] = {'R', 'D', 'F', 'J', 'I', 'T', '1', 0};
bool cache_read(const std::string& path, const char* sig, std::vector<char>& code) {
    std::vector<char> raw;
    if (path.empty() || !read_file(path, raw)) return false;
    const size_t sl = strlen(sig);
    if (raw.size() < 8 + 4 + sl + 8 || memcmp(raw.data(), kCacheMagic, 8) != 0) return false;
    uint32_t n = 0;
    memcpy(&n, raw.data() + 8, 4);
    if (n != sl || memcmp(raw.data() + 12, sig, sl) != 0) return false;           // another program's file under this name: a miss
    uint64_t cl = 0;
    memcpy(&cl, raw.data() + 12 + sl, 8);
    if (cl == 0 || 12 + sl + 8 + cl != raw.size()) return false;
    code.assign(raw.begin() + (long)(12 + sl + 8), raw.end());
    return true;
}
void cache_write(const std::string& path, const char* sig, const std::vector<char>& code) {
    if (path.empty()) return;
    const std::string tmp = path + "." + std::to_string((long)getpid()) + "." + std::to_string((long)(uintptr_t)&code & 0xffff);   // renamed into place: a reader never sees half a file
    FILE* c = std::fopen(tmp.c_str(), "wb");
    if (!c) return;
    const uint32_t n = (uint32_t)strlen(sig);
    const uint64_t cl = code.size();
    const bool w = std::fwrite(kCacheMagic, 1, 8, c) == 8 && std::fwrite(&n, 4, 1, c) == 1 && std::fwrite(sig, 1, n, c) == n && std::fwrite(&cl, 8, 1, c) == 1 &&
                   std::fwrite(code.data(), 1, code.size(), c) == code.size();
    std::fclose(c);
    if (!w || rename(tmp.c_str(), path.c_str()) != 0) (void)unlink(tmp.c_str());
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
        std::vector<std::string> argv_s = {ps.hipcc, arch_opt, "--cuda-device-only", "--no-gpu-bundle-output", "-O3", "-std=c++17", "-ffp-contract=off", inc, "-c", src_path, "-o", obj_path};   // (kJitFlags: part of the cache key)
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
                else if (waited_ms >= limit_ms || g_shutdown.load(std::memory_order_acquire)) {
                    (void)kill(pid, SIGKILL);                              // no orphaned hipcc: the child is reaped before the scratch directory goes
                    while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {}
                    why = waited_ms >= limit_ms ? "the compiler did not finish in time" : "the library is shutting down";
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

// the code object of a signature: from the cache directory, or one run of the compiler.  No HIP call in here: it runs on
// helper threads as well.
bool obtain_code(const char* sig, const std::string& arch, Code& c) {
    const Paths& ps = paths();
    static const bool dbg = getenv("RDF_DEBUG_JIT") != nullptr;
    std::string type;
    c.grouped = sig[0] == 'G';
    if (!(c.grouped ? gprog_type(sig, type) : prog_type(sig, type))) { c.why = "not an exact-program signature"; return false; }
    const std::string cached = ps.ok || !ps.stamp.empty() ? cache_file(sig, arch, ps) : std::string();
    bool from_cache = false;
    if (cache_read(cached, sig, c.bytes)) {
        from_cache = true;
        if (dbg) fprintf(stderr, "[rdf] jit: %s from %s\n", sig, cached.c_str());
    } else {
        if (!ps.ok) { c.why = ps.why; return false; }
        if (!compile(ps, "--offload-arch=" + arch, type, c.grouped, c.bytes, c.why)) return false;
        cache_write(cached, sig, c.bytes);
    }
    if (!kernel_symbol(c.bytes, c.grouped ? "_ZN4rdfk12gspec_kernel" : "_ZN4rdfk11spec_kernel", c.kname)) { c.why = "kernel symbol not found in the code object"; return false; }
    if (dbg) fprintf(stderr, "[rdf] jit: %zu bytes of code, loading %s\n", c.bytes.size(), c.kname.c_str());
    std::lock_guard<std::mutex> lock(g_mu);
    if (from_cache) ++g_from_cache; else ++g_compiled;
    return true;
}

// load a ready code object on the CURRENT device and read its tile size
bool load_here(const Code& c, Loaded& l, std::string& why) {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr, meta = nullptr;
    if (hipModuleLoadData(&mod, c.bytes.data()) != hipSuccess) { (void)hipGetLastError(); why = "hipModuleLoadData failed"; return false; }
    bool ok = false;
    do {
        if (hipModuleGetFunction(&fn, mod, c.kname.c_str()) != hipSuccess || hipModuleGetFunction(&meta, mod, "rdf_jit_meta") != hipSuccess) { why = "kernel not found in the code object"; break; }
        int* d_meta = nullptr;
        int h_meta[4] = {0, 0, 0, 0};
        if (hipMalloc((void**)&d_meta, 16) != hipSuccess) { why = "hipMalloc failed"; break; }
        void* margs[] = {(void*)&d_meta};
        const bool launched = hipModuleLaunchKernel(meta, 1, 1, 1, 1, 1, 1, 0, nullptr, margs, nullptr) == hipSuccess &&
                              hipMemcpy(h_meta, d_meta, 16, hipMemcpyDeviceToHost) == hipSuccess;
        (void)hipFree(d_meta);
        if (!launched || h_meta[0] <= 0) { why = "the kernel's tile size could not be read"; break; }
        l.k.fn = (void*)fn;
        l.k.rows_per_tile = c.grouped ? kEvalTile : 64 * h_meta[0];
        l.k.nvalues = c.grouped ? h_meta[1] : 0;
        ok = true;
    } while (false);
    if (!ok) { (void)hipGetLastError(); (void)hipModuleUnload(mod); }
    return ok;
}

// (g_mu held) the kernel of `sig` on device `dev`, loading the ready code object there on first use
const JitKernel* find_locked(int dev, const char* sig) {
    auto key = std::make_pair(dev, std::string(sig));
    auto it = g_loaded.find(key);
    if (it != g_loaded.end()) return it->second.failed ? nullptr : &it->second.k;
    auto ci = g_code.find(sig);
    if (ci == g_code.end() || ci->second->state != kReady) return nullptr;
    Loaded l;
    std::string why;
    l.failed = !load_here(*ci->second, l, why);
    static const bool dbg = getenv("RDF_DEBUG") != nullptr || getenv("RDF_DEBUG_JIT") != nullptr;
    if (dbg) {
        if (l.failed) fprintf(stderr, "[rdf] jit: %s on device %d -> interpreter (%s)\n", sig, dev, why.c_str());
        else fprintf(stderr, "[rdf] jit: compiled spec_kernel<%s>, %d rows per wave iteration (device %d)\n", sig, l.k.rows_per_tile, dev);
    }
    auto ins = g_loaded.emplace(key, l);
    return l.failed ? nullptr : &ins.first->second.k;
}

void finish_code(const std::string& sig, const std::string& arch, std::shared_ptr<Code> c) {
    bool run = true;
    {
        std::unique_lock<std::mutex> lock(g_mu);
        g_cv.wait(lock, [] { return g_running < kMaxCompilers || g_shutdown.load(std::memory_order_acquire); });
        run = !g_shutdown.load(std::memory_order_acquire);
        ++g_running;
    }
    if (!run) c->why = "the library is shutting down";
    const bool ok = run && obtain_code(sig.c_str(), arch, *c);
    static const bool dbg = getenv("RDF_DEBUG") != nullptr || getenv("RDF_DEBUG_JIT") != nullptr;
    if (!ok && dbg) fprintf(stderr, "[rdf] jit: %s -> interpreter (%s)\n", sig.c_str(), c->why.c_str());
    std::lock_guard<std::mutex> lock(g_mu);
    --g_running;
    if (!ok) ++g_failed;
    c->state = ok ? kReady : kFailed;
    g_cv.notify_all();
}

void helper_main(std::string sig, std::string arch, std::shared_ptr<Code> c, std::shared_ptr<std::atomic<bool>> done) {
    finish_code(sig, arch, c);
    done->store(true, std::memory_order_release);
}
// (g_mu NOT held) start a helper for `sig`; helpers that have finished are joined on the way.  false: shutting down.
bool start_helper(const char* sig, const std::string& arch, const std::shared_ptr<Code>& c) {
    std::vector<std::thread> finished;
    bool started = false;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        for (size_t i = 0; i < g_helpers.size();) {
            if (g_helpers[i].done->load(std::memory_order_acquire)) { finished.push_back(std::move(g_helpers[i].t)); g_helpers.erase(g_helpers.begin() + (long)i); }
            else ++i;
        }
        if (!g_shutdown.load(std::memory_order_acquire)) {
            static const bool registered = (atexit(jit_shutdown), true);   // exit() from any thread: before the static destructors of this library run
            (void)registered;
            Helper h;
            h.done = std::make_shared<std::atomic<bool>>(false);
            h.t = std::thread(helper_main, std::string(sig), arch, c, h.done);
            g_helpers.push_back(std::move(h));
            started = true;
        }
    }
    for (std::thread& t : finished) t.join();
    return started;
}

}  // namespace

// End of the process (atexit) or of the library (its destructor, dlclose): compilers in flight are killed, their scratch
// directories removed by the helpers that own them, and every helper thread is joined — nothing of this library runs afterwards.
void jit_shutdown() {
    std::vector<Helper> all;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        g_shutdown.store(true, std::memory_order_release);
        all.swap(g_helpers);
        g_cv.notify_all();
    }
    for (Helper& h : all) if (h.t.joinable()) h.t.join();
}
namespace { __attribute__((destructor)) void jit_library_unload() { jit_shutdown(); } }

const JitKernel* jit_find(const char* sig) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lock(g_mu);
    return find_locked(dev, sig);
}

// wait = false: a shape met for the first time is compiled on a helper thread and THIS call is answered by the interpreter
// (nullptr); the kernel takes over once it is ready.  wait = true: the caller waits for the compiler (tests, warm-up runs).
// A code object found in the cache directory is loaded at once either way.
const JitKernel* jit_spec_kernel(const char* sig, bool wait) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    const std::string arch = prop.gcnArchName;                                  // "gfx950:sramecc+:xnack-"
    std::shared_ptr<Code> c;
    bool mine = false;
    {
        std::unique_lock<std::mutex> lock(g_mu);
        if (const JitKernel* k = find_locked(dev, sig)) return k;
        auto it = g_code.find(sig);
        if (it == g_code.end()) {
            c = std::make_shared<Code>();
            g_code.emplace(sig, c);
            mine = true;
        } else {
            c = it->second;
            if (c->state == kFailed) return nullptr;
            if (c->state == kReady) return nullptr;                             // ready but not loadable on this device (remembered in g_loaded)
            if (!wait) return nullptr;                                          // somebody is compiling it: interpret this once more
            g_cv.wait(lock, [&] { return c->state != kCompiling; });
            return find_locked(dev, sig);
        }
    }
    // a cached code object answers at once (no compiler run): look before deciding who waits
    if (mine) {
        const Paths& ps = paths();
        std::vector<char> probe;
        const bool cached = cache_read((ps.ok || !ps.stamp.empty()) ? cache_file(sig, arch, ps) : std::string(), sig, probe);
        if (wait || cached) finish_code(sig, arch, c);
        else {
            if (!start_helper(sig, arch, c)) {                                  // shutting down: nobody will compile it
                std::lock_guard<std::mutex> lock(g_mu);
                c->state = kFailed;
                c->why = "the library is shutting down";
                g_cv.notify_all();
            }
            return nullptr;
        }
    }
    std::lock_guard<std::mutex> lock(g_mu);
    return find_locked(dev, sig);
}

// a launch of a run-time kernel failed: this (device, signature) goes to the interpreter from now on
void jit_mark_failed(const char* sig) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return; }
    std::lock_guard<std::mutex> lock(g_mu);
    g_loaded[std::make_pair(dev, std::string(sig))].failed = true;
}

int jit_compiled_count() { std::lock_guard<std::mutex> lock(g_mu); return g_compiled + g_from_cache; }

// one line for rdf_jit_status(): can shapes outside the catalogs be compiled here, and what has happened so far
std::string jit_status() {
    const Paths& ps = paths();
    const CacheDir& cd = cache_dir();
    std::lock_guard<std::mutex> lock(g_mu);
    int compiling = 0;
    for (auto& kv : g_code) compiling += kv.second->state == kCompiling;
    char counts[160];
    std::snprintf(counts, sizeof counts, "%d compiled, %d from the cache, %d failed, %d in progress", g_compiled, g_from_cache, g_failed, compiling);
    if (!ps.ok)
        return "run-time compiler unavailable: " + ps.why + " — program shapes outside the catalogs run on the interpreter"
               + (cd.dir.empty() ? std::string() : " unless " + cd.dir + " holds their code objects") + " (" + counts + ")";
    return "run-time compiler ready: " + ps.hipcc + ", kernel sources " + ps.src + ", code-object cache " + (cd.dir.empty() ? "off (" + cd.why + ")" : cd.dir) + " (" + counts + ")";
}

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
