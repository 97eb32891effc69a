// User-defined separable energies: LambdaDistribution beyond the built-in families (README.md:27-36 promises that the
// caller's energy_func / energy_grad_func define the distribution, mjhmc/misc/distributions.py:198-251).  A Python
// callable cannot run on the GPU, so the caller states the same functions as C expressions of one coordinate,
//
//     E(x) = sum_d  e(x_d, d; p)        dE/dx_d = g(x_d, d; p)
//
// (`x` the coordinate, `d` its index, `p[k]` float64 parameters; optionally coupled through per-particle statistics
// S[k] = sum_d s_k(x_d, d; p), see user_expr_source), and this file compiles the engine's own kernel templates
// (elementwise.hpp: mjhmc_jump_kernel, mjhmc_eval_kernel, mjhmc_leap_kernel) around them with hipRTC, at energy
// creation, for gfx950.  The resulting energy runs through exactly the code paths of the built-in elementwise
// energies: same lane mapping, same jump / accept logic, same counter RNG, all three sampler families, replay mode.
#include <dlfcn.h>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <string>
#include <vector>

#include "handles.hpp"
#include "user_expr.hpp"

namespace {

struct Rtc {
  void* so = nullptr;
  decltype(&hiprtcCreateProgram) CreateProgram = nullptr;
  decltype(&hiprtcDestroyProgram) DestroyProgram = nullptr;
  decltype(&hiprtcAddNameExpression) AddNameExpression = nullptr;
  decltype(&hiprtcCompileProgram) CompileProgram = nullptr;
  decltype(&hiprtcGetProgramLogSize) GetProgramLogSize = nullptr;
  decltype(&hiprtcGetProgramLog) GetProgramLog = nullptr;
  decltype(&hiprtcGetLoweredName) GetLoweredName = nullptr;
  decltype(&hiprtcGetCodeSize) GetCodeSize = nullptr;
  decltype(&hiprtcGetCode) GetCode = nullptr;
  std::string err;
};

Rtc& rtc() {
  static Rtc r;
  if (r.so || !r.err.empty()) return r;
  // MJHMC_HIPRTC_LIB names the library to load and is then the ONLY name tried; otherwise the sonames of the ROCm install
  const char* named = std::getenv("MJHMC_HIPRTC_LIB");
  const char* names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
  if (named && *named) {
    r.so = dlopen(named, RTLD_NOW | RTLD_LOCAL);
  } else {
    for (const char* n : names) {
      r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
    }
  }
  if (!r.so) {
    const char* why = dlerror();  // ONE call: dlerror() clears the message it returns
    r.err = std::string("libhiprtc.so could not be loaded: ") + (why ? why : "?");
    return r;
  }
  auto sym = [&](const char* name) -> void* {
    void* p = dlsym(r.so, name);
    if (!p && r.err.empty()) r.err = std::string("libhiprtc.so lacks ") + name;
    return p;
  };
  r.CreateProgram = (decltype(r.CreateProgram))sym("hiprtcCreateProgram");
  r.DestroyProgram = (decltype(r.DestroyProgram))sym("hiprtcDestroyProgram");
  r.AddNameExpression = (decltype(r.AddNameExpression))sym("hiprtcAddNameExpression");
  r.CompileProgram = (decltype(r.CompileProgram))sym("hiprtcCompileProgram");
  r.GetProgramLogSize = (decltype(r.GetProgramLogSize))sym("hiprtcGetProgramLogSize");
  r.GetProgramLog = (decltype(r.GetProgramLog))sym("hiprtcGetProgramLog");
  r.GetLoweredName = (decltype(r.GetLoweredName))sym("hiprtcGetLoweredName");
  r.GetCodeSize = (decltype(r.GetCodeSize))sym("hiprtcGetCodeSize");
  r.GetCode = (decltype(r.GetCode))sym("hiprtcGetCode");
  return r;
}

}  // namespace

// `stats`: expressions s_k(x, d; p) separated by ';' (empty: none).  Their per-particle sums S[k] = sum_d s_k(x_d, d) are
// available to the other expressions, which makes the family
//     E(x) = e0(S; p) + sum_d e(x_d, d, S; p),      dE/dx_d = g(x_d, d, S; p)
// -- separable GIVEN a handful of global statistics (Neal's funnel, mean-field couplings, radial energies).  The caller
// supplies g with the chain-rule terms through S written out.
static std::vector<std::string> split_stats(const std::string& stats) {
  std::vector<std::string> out;
  std::string cur;
  for (char ch : stats) {
    if (ch == ';') {
      out.push_back(cur);
      cur.clear();
    } else {
      cur.push_back(ch);
    }
  }
  bool blank = true;
  for (char ch : cur) blank = blank && (ch == ' ' || ch == '\t' || ch == '\n');
  if (!blank) out.push_back(cur);
  return out;
}

std::string user_expr_source(const std::string& energy_expr, const std::string& grad_expr, const std::string& stats,
                             const std::string& energy0_expr) {
  const std::vector<std::string> st = split_stats(stats);
  const int K = (int)st.size();
  const std::string e0 = energy0_expr.empty() ? "0.0" : energy0_expr;
  std::string s;
  s += "#include \"elementwise.hpp\"\n";
  s += "namespace mjhmc {\n";
  s += "// E(x) = e0(S) + sum_d e(x_d, d, S; p), dE/dx_d = g(x_d, d, S; p), S[k] = sum_d s_k(x_d, d; p): the caller's expressions\n";
  s += "struct UserExprF {\n";
  s += "  static constexpr bool kLinearIso = false;\n";
  s += "  static constexpr bool kFuse = true;\n";
  s += "  const double* p;  // parameters, device memory\n";
  s += "  int D;\n";
  s += "  template <int E> using Local = NoLocal<E>;\n";
  s += "  template <int E> __device__ __forceinline__ Local<E> local(const LaneMap&) const { return {}; }\n";
  if (K == 0) {
    s += "  struct Ctx { const double* S = nullptr; };\n";
    s += "  template <int E> __device__ __forceinline__ Ctx prep(const double (&)[E], const LaneMap&) const { return {}; }\n";
  } else {
    s += "  struct Ctx { double S[" + std::to_string(K) + "]; };\n";
    for (int k = 0; k < K; ++k)
      s += "  __device__ __forceinline__ double s" + std::to_string(k) + "_of(double x, int d) const { (void)d; (void)x; return (double)(" +
           st[(size_t)k] + "); }\n";
    s += "  template <int E> __device__ __forceinline__ Ctx prep(const double (&x)[E], const LaneMap& m) const {\n";
    s += "    Ctx c;\n";
    for (int k = 0; k < K; ++k) {
      const std::string ks = std::to_string(k);
      s += "    { double a = 0.0;\n";
      s += "#pragma unroll\n";
      s += "      for (int e = 0; e < E; ++e) { const int d = dim_of<double, E>(m, e); a += d < m.D ? s" + ks + "_of(x[e], d) : 0.0; }\n";
      s += "      c.S[" + ks + "] = group_sum(a, m.G); }\n";
    }
    s += "    return c;\n";
    s += "  }\n";
  }
  s += "  __device__ __forceinline__ double e_of(double x, int d, const double* S) const { (void)d; (void)S; return (double)(" + energy_expr + "); }\n";
  s += "  __device__ __forceinline__ double g_of(double x, int d, const double* S) const { (void)d; (void)S; return (double)(" + grad_expr + "); }\n";
  s += "  __device__ __forceinline__ double e0_of(const double* S) const { (void)S; return (double)(" + e0 + "); }\n";
  s += "  template <int E> __device__ __forceinline__ double grad(double xe, int, int d, const Ctx& c, const Local<E>&) const {\n";
  s += "    return d < D ? g_of(xe, d, c.S) : 0.0;  // padded elements keep x = v = 0\n";
  s += "  }\n";
  s += "  template <int E> __device__ __forceinline__ double energy(const double (&x)[E], const LaneMap& m, const Local<E>&) const {\n";
  s += "    const Ctx c = prep<E>(x, m);\n";
  s += "    double s = 0.0;\n";
  s += "#pragma unroll\n";
  s += "    for (int e = 0; e < E; ++e) { const int d = dim_of<double, E>(m, e); s += d < m.D ? e_of(x[e], d, c.S) : 0.0; }\n";
  s += "    return group_sum(s, m.G) + e0_of(c.S);\n";
  s += "  }\n";
  s += "};\n";
  s += "}  // namespace mjhmc\n";
  return s;
}

std::vector<std::string> user_expr_kernel_names(int E) {
  const std::string e = std::to_string(E);
  std::vector<std::string> n;
  for (int mode = 0; mode < 3; ++mode)
    for (int replay = 0; replay < 2; ++replay)
      n.push_back("mjhmc::mjhmc_jump_kernel<mjhmc::UserExprF, double, " + e + ", " + std::to_string(mode) + ", " +
                  (replay ? "true" : "false") + ", false, 0, false>");
  n.push_back("mjhmc::mjhmc_eval_kernel<mjhmc::UserExprF, double, " + e + ">");
  n.push_back("mjhmc::mjhmc_leap_kernel<mjhmc::UserExprF, double, " + e + ">");
  return n;
}

int user_expr_compile(const std::string& src, const std::string& include_dir, int E, std::vector<char>* code,
                      std::vector<std::string>* lowered, std::string* err) {
  Rtc& r = rtc();
  if (!r.err.empty()) {
    *err = r.err;
    return MJHMC_ERR_HIP;
  }
  hiprtcProgram prog;
  if (r.CreateProgram(&prog, src.c_str(), "mjhmc_user_energy.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    *err = "hiprtcCreateProgram failed";
    return MJHMC_ERR_HIP;
  }
  const std::vector<std::string> names = user_expr_kernel_names(E);
  for (const std::string& n : names) r.AddNameExpression(prog, n.c_str());
  const std::string inc = "-I" + include_dir;
  // the library's own build flags (csrc/Makefile): the leapfrog update rounds like the reference's NumPy expression
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-disable-machine-licm",
                        "-DMJHMC_JUMP_WAVES=1", inc.c_str()};
  const hiprtcResult rc = r.CompileProgram(prog, (int)(sizeof(opts) / sizeof(opts[0])), opts);
  if (rc != HIPRTC_SUCCESS) {
    size_t ls = 0;
    r.GetProgramLogSize(prog, &ls);
    std::string log(ls, '\0');
    if (ls) r.GetProgramLog(prog, &log[0]);
    *err = "the energy expressions do not compile:\n" + log;
    r.DestroyProgram(&prog);
    return MJHMC_ERR_INVALID;
  }
  lowered->clear();
  for (const std::string& n : names) {
    const char* low = nullptr;
    if (r.GetLoweredName(prog, n.c_str(), &low) != HIPRTC_SUCCESS || !low) {
      *err = "no lowered name for " + n;
      r.DestroyProgram(&prog);
      return MJHMC_ERR_HIP;
    }
    lowered->push_back(low);
  }
  size_t cs = 0;
  r.GetCodeSize(prog, &cs);
  code->resize(cs);
  r.GetCode(prog, code->data());
  r.DestroyProgram(&prog);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
int user_energy_build(mjhmc_energy* e, const char* energy_expr, const char* grad_expr, const char* stats,
                      const char* energy0_expr, const char* include_dir, const double* params, size_t nparams, int E) {
  UserEnergy* u = new UserEnergy();
  e->user = u;
  u->E = E;
  std::vector<char> code;
  std::vector<std::string> lowered;
  std::string err;
  const int rc = user_expr_compile(user_expr_source(energy_expr, grad_expr, stats ? stats : "", energy0_expr ? energy0_expr : ""),
                                   include_dir, E, &code, &lowered, &err);
  if (rc) return mjhmc_fail(rc, err);
  HIPCHK(hipModuleLoadData(&u->module, code.data()));
  for (int mode = 0; mode < 3; ++mode)
    for (int replay = 0; replay < 2; ++replay)
      HIPCHK(hipModuleGetFunction(&u->jump[mode][replay], u->module, lowered[(size_t)mode * 2 + replay].c_str()));
  HIPCHK(hipModuleGetFunction(&u->eval, u->module, lowered[6].c_str()));
  HIPCHK(hipModuleGetFunction(&u->leap, u->module, lowered[7].c_str()));
  const size_t nb = (nparams ? nparams : 1) * sizeof(double);
  HIPCHK(hipMalloc((void**)&u->dparams, nb));
  if (nparams) HIPCHK(hipMemcpy(u->dparams, params, nparams * sizeof(double), hipMemcpyHostToDevice));
  int cus = 0;
  HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->ctx->device));
  for (int mode = 0; mode < 3; ++mode)
    for (int replay = 0; replay < 2; ++replay) {
      int per_cu = 0;
      HIPCHK(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, u->jump[mode][replay], 256, 0));
      u->resident[mode][replay] = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
    }
  return 0;
}

void user_energy_free(mjhmc_energy* e) {
  if (!e->user) return;
  if (e->user->dparams) (void)hipFree(e->user->dparams);
  if (e->user->module) (void)hipModuleUnload(e->user->module);
  delete e->user;
  e->user = nullptr;
}

// the functor as the RTC translation unit declares it: {const double* p; int D;}
struct UserFunctorArg {
  const double* p;
  int D;
};

int user_launch_jump(const mjhmc_energy* e, const JumpArgs<double>& a, hipStream_t st) {
  const UserEnergy* u = e->user;
  const int replay = a.noise != nullptr ? 1 : 0;
  JumpArgs<double> args = a;
  args.n_fuse = 0;
  UserFunctorArg en{u->dparams, e->ep.ndims};
  void* params[] = {&args, &en};
  const int64_t nslots = a.Npad >> (6 - a.logG);
  const int64_t want = (nslots + 3) / 4;
  const unsigned grid = (unsigned)(want < u->resident[a.mode][replay] ? want : u->resident[a.mode][replay]);
  HIPCHK(hipModuleLaunchKernel(u->jump[a.mode][replay], grid, 1, 1, 256, 1, 1, 0, st, params, nullptr));
  return 0;
}

int user_launch_eval(const mjhmc_energy* e, const EvalArgs<double>& a, hipStream_t st) {
  const UserEnergy* u = e->user;
  EvalArgs<double> args = a;
  UserFunctorArg en{u->dparams, e->ep.ndims};
  void* params[] = {&args, &en};
  const int64_t threads = a.N << a.logG;
  HIPCHK(hipModuleLaunchKernel(u->eval, (unsigned)((threads + 255) / 256), 1, 1, 256, 1, 1, 0, st, params, nullptr));
  return 0;
}

int user_launch_leap(const mjhmc_energy* e, const LeapArgs<double>& a, hipStream_t st) {
  const UserEnergy* u = e->user;
  LeapArgs<double> args = a;
  UserFunctorArg en{u->dparams, e->ep.ndims};
  void* params[] = {&args, &en};
  const int64_t threads = a.N << a.logG;
  HIPCHK(hipModuleLaunchKernel(u->leap, (unsigned)((threads + 255) / 256), 1, 1, 256, 1, 1, 0, st, params, nullptr));
  return 0;
}
