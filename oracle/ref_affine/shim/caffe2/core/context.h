// TEST INFRASTRUCTURE (oracle/): a minimal stand-in for the part of Caffe2's public operator API that
// /root/reference/lib/ops/affine_channel_nd_op.{h,cu} uses, so that the reference's OWN translation unit compiles with hipcc and its
// own RunOnDevice() bodies (shape checks, grid computation, kernel launches) execute.  Caffe2 @ b4e1588 itself is not in the tree
// (SURVEY.md F1/F7); the names and semantics below follow its published headers (caffe2/core/{tensor,operator,common_gpu}.h):
//   Tensor: ndim(), dim32(i), size(), data<T>(), mutable_data<T>(), ResizeLike();   Operator<Context>: Input(i), Output(i), context_;
//   CUDA_1D_KERNEL_LOOP = grid-stride loop over int indices;  CAFFE_CUDA_NUM_THREADS = 512;  CAFFE_MAXIMUM_NUM_BLOCKS = 4096;
//   CAFFE_GET_BLOCKS(N) = min((N + 511) / 512, 4096);  CAFFE_ENFORCE_EQ throws.
// Nothing here is product code; the product never links it (tests/test_host_cpu.py::test_product_never_imports_oracle).
#ifndef DAT_ORACLE_CAFFE2_SHIM_H_
#define DAT_ORACLE_CAFFE2_SHIM_H_
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

namespace caffe2 {

using std::string;
using std::vector;

struct OperatorDef {};
struct Workspace {};
struct CPUContext {};

struct CUDAContext {
  hipStream_t stream = nullptr;
  hipStream_t cuda_stream() const { return stream; }
};

// A tensor that VIEWS caller-owned device memory (the shim never allocates: ResizeLike only adopts the shape).
class Tensor {
 public:
  Tensor() = default;
  Tensor(void* p, std::vector<int> dims) : ptr_(p), dims_(std::move(dims)) {}
  int ndim() const { return (int)dims_.size(); }
  int dim32(int i) const { return dims_.at(i); }
  long long size() const {
    long long n = 1;
    for (int d : dims_) n *= d;
    return n;
  }
  template <typename T>
  const T* data() const { return static_cast<const T*>(ptr_); }
  template <typename T>
  T* mutable_data() { return static_cast<T*>(ptr_); }
  void ResizeLike(const Tensor& o) { dims_ = o.dims_; }

 private:
  void* ptr_ = nullptr;
  std::vector<int> dims_;
};

template <class Context>
class Operator {
 public:
  Operator(const OperatorDef&, Workspace*) {}
  virtual ~Operator() {}
  virtual bool RunOnDevice() = 0;
  const Tensor& Input(int i) { return *inputs_.at(i); }
  Tensor* Output(int i) { return outputs_.at(i); }
  // (test driver side)
  std::vector<const Tensor*> inputs_;
  std::vector<Tensor*> outputs_;
  Context context_;
};

#define USE_OPERATOR_CONTEXT_FUNCTIONS          \
  using Operator<Context>::Input;               \
  using Operator<Context>::Output;              \
  using Operator<Context>::context_

#define CAFFE_NOT_IMPLEMENTED throw std::runtime_error("CAFFE_NOT_IMPLEMENTED")
#define CAFFE_ENFORCE_EQ(a, b)                                                                        \
  do {                                                                                                \
    if (!((a) == (b))) throw std::runtime_error(std::string("CAFFE_ENFORCE_EQ failed: " #a " == " #b)); \
  } while (0)

constexpr int CAFFE_CUDA_NUM_THREADS = 512;
constexpr int CAFFE_MAXIMUM_NUM_BLOCKS = 4096;
inline int CAFFE_GET_BLOCKS(const int N) {
  return std::min((N + CAFFE_CUDA_NUM_THREADS - 1) / CAFFE_CUDA_NUM_THREADS, CAFFE_MAXIMUM_NUM_BLOCKS);
}
#define CUDA_1D_KERNEL_LOOP(i, n) \
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += blockDim.x * gridDim.x)

// operator registration is Caffe2's plugin boundary (replaced by the C ABI here): a no-op declaration
#define DAT_SHIM_CAT2(a, b) a##b
#define DAT_SHIM_CAT(a, b) DAT_SHIM_CAT2(a, b)
#define REGISTER_CUDA_OPERATOR(name, ...) static const int DAT_SHIM_CAT(dat_shim_reg_##name##_, __LINE__) = 0
#define REGISTER_CPU_OPERATOR(name, ...) static const int DAT_SHIM_CAT(dat_shim_cpureg_##name##_, __LINE__) = 0

}  // namespace caffe2
#endif
