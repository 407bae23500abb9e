// oracle/_ref harness -- TEST INFRASTRUCTURE, built only where /root/reference exists.
//
// Compiles the reference's own CPU arithmetic, from the sources where they lie, into
// oracle/_ref/libscn_ref.so so that the numpy oracle (oracle/arith.py) can be pinned against it:
//   * BatchNormalization_ForwardPass / _BackwardPass   sparseconvnet/SCN/CPU/BatchNormalization.cpp:12-107
//   * InputLayer_ForwardPass / _BackwardPass           sparseconvnet/SCN/CPU/IOLayers.cpp:12-47
//   * rule_index_select / rule_index_add_              sparseconvnet/SCN/CPU/Convolution.cpp:8-33
// These are the complete arithmetic of the hot path; the drivers around them
// (cpu_*Convolution_updateOutput etc.) need the reference's Metadata<D>, whose header includes
// CUDA/CUDPP headers this image lacks (sparseconvnet/SCN/Metadata/Metadata.h:59), so they are NOT
// instantiated: the per-offset loop "select rows -> at::mm -> add rows" is restated below with the
// reference's helpers (it is the loop body of sparseconvnet/SCN/CPU/Convolution.cpp:60-68).
// No reference source text is copied into this repository; nothing here stands in for CUDA/CUDPP.
#include <ATen/ATen.h>
#include <ATen/Parallel.h>
#include <omp.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "sparseconvnet/SCN/Metadata/32bits.h"  // `Int`, Point<> helpers (plain ATen, no CUDA)

// The reference declares these in Metadata.h / defines it in Metadata.cpp:28-30; only the
// declaration is needed for the CPU/*.cpp templates below to parse.
template <Int dimension> class Metadata;
template <typename T> T *OptionalTensorData(at::Tensor tensor) { return tensor.numel() ? tensor.data_ptr<T>() : nullptr; }

#include "sparseconvnet/SCN/CPU/BatchNormalization.cpp"
#include "sparseconvnet/SCN/CPU/Convolution.cpp"
#include "sparseconvnet/SCN/CPU/IOLayers.cpp"

template <typename T> static at::Tensor wrap(T *p, int64_t r, int64_t c) {
  return at::from_blob(p, {r, c}, at::TensorOptions().dtype(c10::CppTypeToScalarType<T>::value));
}

extern "C" {
// threads of the reference's `#pragma omp parallel for` loops (CPU/Convolution.cpp:15,26, CPU/IOLayers.cpp:16) and of at::mm
void ref_set_threads(int n) {
  omp_set_num_threads(n);
  at::set_num_threads(n);
}
}

// Every entry point exists twice: T = float (the arithmetic the reference runs; suffix none) and T = double (suffix _f64) --
// the SAME templates of the reference instantiated at double precision, i.e. the reference's own loops without their float32
// rounding.  At 600k rows the float32 loops' sequential sums (BatchNormalization.cpp:24-36, 75-87) are themselves ~1e-3 away
// from that; the float64 instantiation is what a 1e-4 comparison can be made against (bench.py parity_fullsize).  (One float
// operation survives in it: invstd is computed with powf in every instantiation, BatchNormalization.cpp:40 -- ~6e-8 per batch norm.)
#define SCN_REF_ENTRY_POINTS(T, SUFFIX)                                                                                          \
  extern "C" {                                                                                                                   \
  void ref_bn_forward##SUFFIX(T *x, T *y, int32_t nPlanes, int32_t nActive, T *saveMean, T *saveInvStd, T *runningMean,          \
                              T *runningVar, T *weight, T *bias, T eps, T momentum, int train, T leakiness) {                    \
    BatchNormalization_ForwardPass<T>(x, y, nPlanes, nPlanes, nPlanes, nActive, saveMean, saveInvStd, runningMean, runningVar,  \
                                      weight, bias, eps, momentum, train != 0, leakiness);                                      \
  }                                                                                                                              \
  void ref_bn_backward##SUFFIX(T *x, T *dx, T *y, T *dy, int32_t nPlanes, int32_t nActive, T *saveMean, T *saveInvStd,           \
                               T *runningMean, T *runningVar, T *weight, T *bias, T *d_weight, T *d_bias, T leakiness) {         \
    BatchNormalization_BackwardPass<T>(x, dx, y, dy, nPlanes, nPlanes, nPlanes, nActive, saveMean, saveInvStd, runningMean,     \
                                       runningVar, weight, bias, d_weight, d_bias, leakiness);                                  \
  }                                                                                                                              \
  /* out must be zeroed by the caller ("Assume output and d_input_features have been zero-ed", IOLayers.cpp:9) */               \
  void ref_input_forward##SUFFIX(T *in, T *out, int32_t nRows, int32_t maxActive, int32_t nPlanes, int32_t *rules,               \
                                 int average) {                                                                                  \
    InputLayer_ForwardPass<T>(in, out, nRows, maxActive, nPlanes, rules, average != 0);                                          \
  }                                                                                                                              \
  void ref_input_backward##SUFFIX(T *d_in, T *d_out, int32_t nRows, int32_t maxActive, int32_t nPlanes, int32_t *rules,          \
                                  int average) {                                                                                 \
    InputLayer_BackwardPass<T>(d_in, d_out, nRows, maxActive, nPlanes, rules, average != 0);                                     \
  }                                                                                                                              \
  /* One kernel offset of the gather -> mm -> scatter: out[rules[2i+out_col]] += in[rules[2i+in_col]] . w (w is [Ci,Co]);       \
     the three calls are the body of sparseconvnet/SCN/CPU/Convolution.cpp:60-68. */                                            \
  void ref_rule_conv_offset##SUFFIX(T *in, int64_t n_in, T *out, int64_t n_out, T *w, int32_t Ci, int32_t Co, int32_t *rules,    \
                                    int32_t nRules, int in_col) {                                                                \
    if (!nRules) return;                                                                                                         \
    auto input = wrap<T>(in, n_in, Ci), output = wrap<T>(out, n_out, Co), weight = wrap<T>(w, Ci, Co);                           \
    auto input_rows = rule_index_select<T>(input, nRules, rules + in_col);                                                       \
    auto output_rows = at::mm(input_rows, weight);                                                                               \
    rule_index_add_<T>(output, output_rows, nRules, rules + (1 - in_col));                                                       \
  }                                                                                                                              \
  /* backward of one offset: dW = in_rows^T . dout_rows ; d_in[in idx] += dout_rows . w^T                                       \
     (body of sparseconvnet/SCN/CPU/Convolution.cpp:104-109) */                                                                 \
  void ref_rule_conv_offset_backward##SUFFIX(T *in, T *d_in, int64_t n_in, T *d_out, int64_t n_out, T *w, T *dw, int32_t Ci,     \
                                             int32_t Co, int32_t *rules, int32_t nRules, int in_col) {                           \
    if (!nRules) return;                                                                                                         \
    auto input = wrap<T>(in, n_in, Ci), dinput = wrap<T>(d_in, n_in, Ci), doutput = wrap<T>(d_out, n_out, Co);                   \
    auto weight = wrap<T>(w, Ci, Co), dweight = wrap<T>(dw, Ci, Co);                                                             \
    auto input_rows = rule_index_select<T>(input, nRules, rules + in_col);                                                       \
    auto d_output_rows = rule_index_select<T>(doutput, nRules, rules + (1 - in_col));                                            \
    at::mm_out(dweight, input_rows.t(), d_output_rows);                                                                          \
    auto d_input_rows = at::mm(d_output_rows, weight.t());                                                                       \
    rule_index_add_<T>(dinput, d_input_rows, nRules, rules + in_col);                                                            \
  }                                                                                                                              \
  }

SCN_REF_ENTRY_POINTS(float, )
SCN_REF_ENTRY_POINTS(double, _f64)
