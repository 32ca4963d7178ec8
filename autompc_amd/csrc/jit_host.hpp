// jit_host.hpp -- run-time specialisation of the kernels for the staged model's shape (api.cpp only).
//
// The reference's MLP configuration space is 1-4 hidden layers of 16-256 units (mlp.py:113-122); the
// library pads every hidden layer of a model to one width hpad in {64, 128, 192, 256}, so for a
// given system there are 16 kernel shapes (n_hidden x hpad).  shapes.hpp registers the benchmark
// systems' default networks; every other shape a tuner produces is compiled here on first use:
//
//   ampc_set_mlp      shape not registered -> look for  <cache>/shape_<key>_<srchash>.so ; if it is
//                     absent, start  /bin/sh -c "hipcc ... (4 units in parallel) && link && mv"
//                     in the background (posix_spawn) and return at once
//   plan creation     plugin present (or just finished) -> dlopen, check ampc_jit_info against the
//                     model, plan->jit = plugin, plan->static_shape = 0;  otherwise the DynShape
//                     kernels run (1.1-1.9x slower) and the next plan looks again
//   ampc_jit_status   reaps a finished build: a controller that polls it switches over by itself
//   ampc_jit_wait     block until the build of the handle's shape has finished (tools / tests)
//
// <cache> = $AMPC_JIT_CACHE or <package>/jit_cache (in-tree: it travels with the package; ~/.cache/autompc_amd when
// the package directory is read-only).  The key
// holds precision + shape, the file name also a hash of every source the plugin is compiled from
// and of `hipcc --version`, so a plugin is never used with other headers or another compiler /
// ROCm release than the ones at hand.  Processes that want the same plugin at the same time (the
// ranks of a torch.distributed job) share ONE build through a lock directory in the cache.
// AMPC_JIT=0 disables all of it.  Same arithmetic in the same order as DynShape: results are
// bit-identical (tests/test_gpu_jit.py).
#pragma once
#include <dlfcn.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <signal.h>

#include <fstream>
#include <map>
#include <mutex>
#include <sstream>

#include "host_common.hpp"

extern char** environ;

namespace jit {

struct Entry {
  int state = 0;            // 0 unknown, 1 building, 2 ready, -1 failed
  pid_t pid = -1;
  bool reaped = false;       // the build's process has been waited for (or is not ours to wait for)
  std::string so, log;
  JitPlugin plug;
  int info[8] = {0};
};

inline std::mutex& mu() { static std::mutex m; return m; }
inline std::map<std::string, Entry>& table() { static std::map<std::string, Entry> t; return t; }

inline std::string package_dir() {          // directory of libautompc_hip.so
  Dl_info di;
  if (dladdr((const void*)&package_dir, &di) == 0 || !di.dli_fname) return ".";
  std::string p = di.dli_fname;
  const size_t k = p.rfind('/');
  return k == std::string::npos ? "." : p.substr(0, k);
}

inline const std::vector<std::string>& source_files() {
  static const std::vector<std::string> f = {
      "host_common.hpp", "shapes.hpp", "probe.hpp", "mlp_tile.hpp", "mlp_kernels.hpp", "linear_kernels.hpp", "mppi_kernels.hpp", "mppi_rollout4.hpp",
      "ilqr_kernels.hpp", "ilqr_ls4.hpp", "ilqr_lsw.hpp", "ilqr_wide.hpp", "rng_kernels.hpp", "sindy_kernels.hpp", "score_kernels.hpp",
      "launch_mppi.cpp", "launch_mlp.cpp", "launch_ilqr.cpp", "jit_plugin.cpp", "../../include/autompc_hip.h"};
  return f;
}

// FNV-1a over the plugin's sources: "" if one of them is missing (a package shipped without csrc/)
inline const std::string& source_hash() {
  static const std::string h = [] {
    uint64_t x = 1469598103934665603ull;
    const std::string dir = package_dir() + "/csrc/";
    for (const std::string& f : source_files()) {
      std::ifstream in(dir + f, std::ios::binary);
      if (!in) return std::string();
      char buf[1 << 14];
      while (in.read(buf, sizeof(buf)) || in.gcount() > 0) {
        for (std::streamsize i = 0; i < in.gcount(); ++i) { x ^= (unsigned char)buf[i]; x *= 1099511628211ull; }
      }
    }
    char out[24];
    std::snprintf(out, sizeof(out), "%016llx", (unsigned long long)x);
    return std::string(out);
  }();
  return h;
}

// $AMPC_JIT_CACHE, else <package>/jit_cache (in-tree: it travels with the package), else -- a package installed
// read-only -- $XDG_CACHE_HOME/autompc_amd or ~/.cache/autompc_amd
inline std::string cache_dir() {
  const char* e = std::getenv("AMPC_JIT_CACHE");
  if (e && *e) return std::string(e);
  static const std::string dir = [] {
    const std::string in_tree = package_dir() + "/jit_cache";
    ::mkdir(in_tree.c_str(), 0755);
    if (::access(in_tree.c_str(), W_OK | X_OK) == 0) return in_tree;
    const char* x = std::getenv("XDG_CACHE_HOME");
    const char* home = std::getenv("HOME");
    std::string base = x && *x ? std::string(x) : (home && *home ? std::string(home) + "/.cache" : std::string("/tmp"));
    ::mkdir(base.c_str(), 0755);
    base += "/autompc_amd";
    ::mkdir(base.c_str(), 0755);
    return base;
  }();
  return dir;
}

inline bool exists(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }

inline std::string hipcc_path() {
  const char* e = std::getenv("HIPCC");
  if (e && *e && exists(e)) return e;
  return exists("/opt/rocm/bin/hipcc") ? "/opt/rocm/bin/hipcc" : "hipcc";
}

template <typename T> inline std::string key_of(const ampc_handle* h) {
  char k[96];
  std::snprintf(k, sizeof(k), "%s_x%d_u%d_o%d_h%d_w%d", sizeof(T) == 8 ? "f64" : "f32", h->nx, h->nu,
                h->obs_dim, h->n_hidden, h->hpad);
  return k;
}

// shapes the StaticShape kernels exist for (what the registered ones are used with)
inline bool eligible(const ampc_handle* h) {
  return env_int("AMPC_JIT", 1) != 0 && h->has_mlp && !h->has_sindy &&     // (linear models of <= 32 states are staged as a one-layer identity network: they qualify)
         h->nx >= 1 && h->nx <= 32 && h->obs_dim >= 1 && h->n_hidden >= 1 && h->n_hidden <= kMaxHidden &&
         (h->hpad == 64 || h->hpad == 128 || h->hpad == 192 || h->hpad == 256) && !source_hash().empty();
}

inline bool load(Entry& e, const ampc_handle* h, size_t tsize) {
  void* dl = dlopen(e.so.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!dl) { e.state = -1; e.log = std::string("dlopen: ") + dlerror(); return false; }
  e.plug.dl = dl;
  e.plug.mppi_solve = (int (*)(ampc_mppi_plan*))dlsym(dl, "ampc_jit_mppi_solve");
  e.plug.ilqr_iter = (int (*)(ampc_ilqr_plan*, int))dlsym(dl, "ampc_jit_ilqr_iter");
  e.plug.ilqr_refresh = (int (*)(ampc_ilqr_plan*))dlsym(dl, "ampc_jit_ilqr_refresh");
  e.plug.last_error = (const char* (*)())dlsym(dl, "ampc_jit_last_error");
  auto info = (void (*)(int*))dlsym(dl, "ampc_jit_info");
  if (!e.plug.mppi_solve || !e.plug.ilqr_iter || !e.plug.ilqr_refresh || !e.plug.last_error || !info) {
    e.state = -1; e.log = "plugin lacks an entry point"; return false;
  }
  info(e.info);
  const int want[6] = {h->nx, h->nu, h->obs_dim, h->n_hidden, h->hpad, (int)tsize};
  for (int i = 0; i < 6; ++i)
    if (e.info[i] != want[i]) { e.state = -1; e.log = "plugin was compiled for another shape"; return false; }
  if (e.info[7] != (int)(sizeof(ampc_mppi_plan) + sizeof(ampc_ilqr_plan) + sizeof(ampc_handle))) {
    e.state = -1; e.log = "plugin was compiled against other plan structs"; return false;
  }
  e.state = 2;
  return true;
}

// Identity of the compiler the plugin would be built with: FNV-1a of `hipcc --version` (the ROCm /
// clang version line), part of the plugin's file name -- a cache directory that outlives a ROCm
// upgrade never hands a code object built by the old compiler to the new runtime.
inline const std::string& compiler_id() {
  static const std::string id = [] {
    uint64_t x = 1469598103934665603ull;
    int fd[2];
    if (pipe(fd) == 0) {                                // `hipcc --version`, no shell in between
      posix_spawn_file_actions_t fa;
      posix_spawn_file_actions_init(&fa);
      posix_spawn_file_actions_adddup2(&fa, fd[1], 1);
      posix_spawn_file_actions_addclose(&fa, fd[0]);
      const std::string cc = hipcc_path();
      const char* argv[] = {cc.c_str(), "--version", nullptr};
      pid_t pid = -1;
      const int rc = posix_spawnp(&pid, cc.c_str(), &fa, nullptr, (char* const*)argv, environ);
      posix_spawn_file_actions_destroy(&fa);
      close(fd[1]);
      if (rc == 0) {
        char buf[512];
        ssize_t n;
        while ((n = read(fd[0], buf, sizeof(buf))) > 0)
          for (ssize_t i = 0; i < n; ++i) { x ^= (unsigned char)buf[i]; x *= 1099511628211ull; }
        int st = 0;
        (void)waitpid(pid, &st, 0);
      }
      close(fd[0]);
    }
    char out[16];
    std::snprintf(out, sizeof(out), "%08x", (unsigned)(x ^ (x >> 32)));
    return std::string(out);
  }();
  return id;
}

// The build script.  Every path reaches it through the ENVIRONMENT (AMPC_J_*), never pasted into the
// text, so quotes or spaces in $AMPC_JIT_CACHE / the package path cannot break it.  One builder per
// plugin file across processes (eight torch.distributed ranks staging the same model): the lock is a
// directory next to the plugin (mkdir is atomic) holding the builder's pid; the others wait for the
// plugin to appear, and take over if the builder has died.  A failed build leaves <plugin>.failed.
inline const char* build_script() {
  return R"SH(exec > "$AMPC_J_LOG" 2>&1
LOCK="$AMPC_J_SO.lock"
mine=0
take() { mkdir "$LOCK" 2>/dev/null && echo $$ > "$LOCK/pid" && mine=1; }
trap '[ $mine = 1 ] && rm -rf "$LOCK"' EXIT
take
i=0
while [ $mine = 0 ] && [ ! -f "$AMPC_J_SO" ] && [ ! -f "$AMPC_J_SO.failed" ] && [ $i -lt 3000 ]; do
  p=$(cat "$LOCK/pid" 2>/dev/null)
  if [ -n "$p" ] && ! kill -0 "$p" 2>/dev/null; then rm -rf "$LOCK"; fi     # the builder died: take over
  take || { sleep 0.1; i=$((i+1)); }
done
[ -f "$AMPC_J_SO" ] && exit 0
[ $mine = 1 ] || exit 4
rm -f "$AMPC_J_SO.failed"
mkdir -p "$AMPC_J_TMP" && cd "$AMPC_J_TMP" || { touch "$AMPC_J_SO.failed"; exit 1; }
for u in launch_mppi launch_mlp launch_ilqr jit_plugin; do
  "$AMPC_J_CC" -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-pass-failed \
    -I "$AMPC_J_INC" $AMPC_J_DEFS -c "$AMPC_J_SRC/$u.cpp" -o $u.o &
done
wait
if [ -f launch_mppi.o -a -f launch_mlp.o -a -f launch_ilqr.o -a -f jit_plugin.o ] &&
   "$AMPC_J_CC" --offload-arch=gfx950 -shared -fPIC launch_mppi.o launch_mlp.o launch_ilqr.o jit_plugin.o -o plugin.so &&
   mv plugin.so "$AMPC_J_SO"; then
  cd / && rm -rf "$AMPC_J_TMP"
  exit 0
fi
touch "$AMPC_J_SO.failed"
exit 2
)SH";
}

template <typename T> inline void start_build(Entry& e, const ampc_handle* h, const std::string& key) {
  const std::string dir = cache_dir(), src = package_dir() + "/csrc", inc = package_dir() + "/../include";
  ::mkdir(dir.c_str(), 0755);
  const std::string tmp = dir + "/build_" + key + "_" + std::to_string((long)getpid());
  std::ostringstream defs;         // (no paths in here: plain -D words, split by the shell on purpose)
  defs << "-DAMPC_JIT_PLUGIN -DAMPC_T=" << (sizeof(T) == 8 ? "double -DAMPC_T_IS_F64=1" : "float")
       << " -DAMPC_JIT_NX=" << h->nx << " -DAMPC_JIT_NU=" << h->nu << " -DAMPC_JIT_NO=" << h->obs_dim
       << " -DAMPC_JIT_NH=" << h->n_hidden << " -DAMPC_JIT_HPAD=" << h->hpad;
  e.log = dir + "/shape_" + key + "_" + std::to_string((long)getpid()) + ".log";
  std::vector<std::string> env;
  for (char** p = environ; p && *p; ++p) env.emplace_back(*p);
  env.push_back("AMPC_J_LOG=" + e.log);
  env.push_back("AMPC_J_SO=" + e.so);
  env.push_back("AMPC_J_TMP=" + tmp);
  env.push_back("AMPC_J_SRC=" + src);
  env.push_back("AMPC_J_INC=" + inc);
  env.push_back("AMPC_J_CC=" + hipcc_path());
  env.push_back("AMPC_J_DEFS=" + defs.str());
  std::vector<char*> envp;
  for (std::string& v : env) envp.push_back(&v[0]);
  envp.push_back(nullptr);
  const char* argv[] = {"/bin/sh", "-c", build_script(), nullptr};
  pid_t pid = -1;
  if (posix_spawn(&pid, "/bin/sh", nullptr, nullptr, (char* const*)argv, envp.data()) != 0) {
    e.state = -1; e.log = "posix_spawn(/bin/sh) failed"; return;
  }
  e.pid = pid;
  e.state = 1;
}

// Reap a finished build (never blocks).  The child may not be ours to wait for any more -- the
// process forked after the spawn, or runs with SIGCHLD ignored: waitpid then fails with ECHILD -- so
// the verdict always comes from the files the script leaves: the plugin, or <plugin>.failed.
inline bool child_gone(Entry& e) {
  if (e.reaped) return true;
  int st = 0;
  const pid_t r = waitpid(e.pid, &st, WNOHANG);
  if (r == 0) return false;                             // still running
  if (r < 0 && !exists(e.so) && !exists(e.so + ".failed") && ::kill(e.pid, 0) == 0)
    return false;                                       // not our child, but alive: keep waiting
  e.reaped = true;
  return true;
}

inline void poll(Entry& e, const ampc_handle* h, size_t tsize) {
  if (e.state != 1 || !child_gone(e)) return;
  if (exists(e.so)) { load(e, h, tsize); return; }
  e.state = -1;
  e.log = "build failed, see " + e.log;
}

// Finished builds of OTHER shapes (a model that was staged and dropped before its plugin was ready)
// are reaped whenever anybody asks about any shape, so no build stays a zombie for the life of the
// process; their plugins are loaded when their own shape is next asked for.
inline void reap_all() {
  for (auto& kv : table())
    if (kv.second.state == 1) (void)child_gone(kv.second);
}

// plugin for the handle's staged shape, or nullptr (not eligible / still building / failed).
// block: wait for a running build -- polling with the table's mutex RELEASED in between, so other
// threads keep creating plans and staging handles meanwhile.
template <typename T> inline const JitPlugin* get(const ampc_handle* h, bool block = false) {
  if (!eligible(h)) return nullptr;
  const std::string key = key_of<T>(h);
  for (;;) {
    {
      std::lock_guard<std::mutex> g(mu());
      Entry& e = table()[key];
      if (e.state == 0) {
        e.so = cache_dir() + "/shape_" + key + "_" + source_hash() + "_" + compiler_id() + ".so";
        if (exists(e.so)) load(e, h, sizeof(T));
        else if (h->jit_build) start_build<T>(e, h, key);
        else { table().erase(key); return nullptr; }     // (ampc_handle_set_jit(h, 0): uses, never builds)
      }
      poll(e, h, sizeof(T));
      reap_all();
      if (e.state != 1 || !block || !h->jit_build) return e.state == 2 ? &e.plug : nullptr;
    }
    usleep(20000);
  }
}

// state of the handle's shape; reaps a finished build, so a long-lived controller that only ever asks
// ampc_jit_status (the drop-in classes do, once per run()) sees 1 -> 2 and switches over
template <typename T> inline int status(const ampc_handle* h, std::string* msg) {
  if (!eligible(h)) return 0;
  std::lock_guard<std::mutex> g(mu());
  auto it = table().find(key_of<T>(h));
  if (it == table().end()) return 0;
  poll(it->second, h, sizeof(T));
  reap_all();
  if (msg) *msg = it->second.state == 2 ? it->second.so : it->second.log;
  return it->second.state;
}

}  // namespace jit
