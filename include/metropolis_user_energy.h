/*
 * metropolis_user_energy.h -- contract for a user-written device energy (me_energy_kind ME_ENERGY_USER /
 * ME_ENERGY_USER_INDIRECT).
 *
 * The reference couples the sampler to the physics through a Python callable
 *     energy(real_params, complex_params) -> float            (metropolisengine/metropolis_engine.py:20, :250)
 * The GPU counterpart is one HIP source file that defines
 *
 *     template <typename R>
 *     __device__ R me_user_energy(const R *x, const R *coef);
 *
 * where x[ME_NR + 2*ME_NC] is the chain's state [real params | Re z | Im z] (ME_NR / ME_NC are compile-time macros
 * equal to n_real / n_complex) and coef points to the engine's energy coefficients in device memory (the doubles
 * passed as me_config.energy_coeffs, converted to R; NULL if there are none).  The function must be pure and must
 * not synchronise.  It is compiled around the engine's own kernels into a plugin library:
 *
 *     python -c "from metropolisengine_amd import build; build.build_user_energy('my_energy.h', 'mine', 2, 7)"
 *     -> metropolisengine_amd/lib/libme_user_mine_2_7.so
 *
 * which registers itself when loaded with me_load_plugin(path); engines select it with
 * me_config.energy_kind = ME_ENERGY_USER (inlined call) or ME_ENERGY_USER_INDIRECT (call through a __device__
 * function pointer held in the plugin's code object) and me_config.user_energy_name = "mine".
 * The hard-wall predicate (reject_condition, metropolis_engine.py:142-146, evaluated before the energy, :247) is either
 * the built-in me_config.reject_kind = ME_REJECT_ABS_REAL0_GE or, with ME_REJECT_USER, a second device function the same
 * source file defines after  #define ME_USER_HAS_REJECT :
 *
 *     template <typename R>
 *     __device__ bool me_user_reject(const R *x, const R *coef);      // true = reject the proposal
 *
 * Energy dictionaries (metropolis_engine.py:111-116: {"complex": {term: fn}, "real": {...}, "all": {...}}): instead
 * of me_user_energy the source may define its energy term by term,
 *
 *     #define ME_USER_N_TERMS 2
 *     constexpr unsigned me_user_term_groups(int term);   // bit 0: real-group moves change it, bit 1: complex-group
 *     template <typename R>
 *     __device__ R me_user_energy_term(int term, const R *x, const R *coef);
 *
 * The engine then keeps one energy row per term (ME_FIELD_ENERGY has ME_USER_N_TERMS components); step_real_group /
 * step_complex_group re-evaluate and compare only their group's terms (:214-221, :230-237), step_all all of them;
 * the total is the sum of the terms (:158-162).  Example: examples/user_energy_landau_terms.h.
 */
#ifndef METROPOLIS_USER_ENERGY_H
#define METROPOLIS_USER_ENERGY_H
#include <hip/hip_runtime.h>
/* One explicit fused multiply-add, a * b + c.  hipcc fuses multiply-adds by itself, but in a sum of TWO products
 * (a * b + c * d, e.g. |z|^2 = re * re + im * im) it may fuse either one, and which one can depend on the kernel the
 * function is inlined into.  Energies whose values must be bit-identical in every kernel of the engine (me_step against
 * me_cycle, float32) spell such sums me_fma(a, b, c * d). */
__device__ __forceinline__ float me_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double me_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
#endif
