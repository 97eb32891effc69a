// User-expression energies compiled with hipRTC (user_expr.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "elementwise.hpp"

struct mjhmc_energy;

struct UserEnergy {
  hipModule_t module = nullptr;
  hipFunction_t jump[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};  // [mode][replay]
  int resident[3][2] = {{1, 1}, {1, 1}, {1, 1}};  // persistent grid: resident 256-thread blocks on the device
  hipFunction_t eval = nullptr, leap = nullptr;
  double* dparams = nullptr;
  int E = 0;  // elements per lane the kernels were instantiated for
};

// the translation unit handed to hipRTC, and the kernel instantiations requested from it (6 jump kernels, eval, leap)
std::string user_expr_source(const std::string& energy_expr, const std::string& grad_expr, const std::string& stats = "",
                             const std::string& energy0_expr = "");
std::vector<std::string> user_expr_kernel_names(int E);
int user_expr_compile(const std::string& src, const std::string& include_dir, int E, std::vector<char>* code,
                      std::vector<std::string>* lowered, std::string* err);

int user_energy_build(mjhmc_energy* e, const char* energy_expr, const char* grad_expr, const char* stats,
                      const char* energy0_expr, const char* include_dir, const double* params, size_t nparams, int E);
void user_energy_free(mjhmc_energy* e);
int user_launch_jump(const mjhmc_energy* e, const mjhmc::JumpArgs<double>& a, hipStream_t st);
int user_launch_eval(const mjhmc_energy* e, const mjhmc::EvalArgs<double>& a, hipStream_t st);
int user_launch_leap(const mjhmc_energy* e, const mjhmc::LeapArgs<double>& a, hipStream_t st);
