// TEST INFRASTRUCTURE (oracle/): C entry points over the reference's OWN AffineChannelNd translation unit.
//
// The reference file is compiled WHERE IT LIES (never copied): the #include below pulls /root/reference/lib/ops/affine_channel_nd_op.cu
// -- its two device kernels (ScaleBiasForward / ScaleForward, :20-46) and the two RunOnDevice() bodies (:50-92) -- into this
// translation unit against the Caffe2 stand-in of oracle/ref_affine/shim.  hipcc compiles the CUDA source unchanged (triple-chevron
// launches, blockIdx / blockDim are HIP built-ins too).  Built by oracle/build_ref.py into oracle/_ref/libref_affine.so
// (git-ignored; travels to the GPU box with the snapshot).  tests/test_gpu_kernels.py checks dat_affine_channel_nd_fwd / _bwd and the
// NumPy oracle against it.
#include REF_AFFINE_CU

extern "C" {

// x, scale, bias, y: DEVICE pointers; x / y are N x C x inner fp32 (y may alias x).  Returns 0, or -1 with the CAFFE_ENFORCE text
// in `err` (the reference op throws).  Synchronous.
int ref_affine_channel_nd_fwd(const float* x, const float* scale, const float* bias, float* y, int n, int c, int inner, int scale_len,
                              char* err, int err_len) {
  try {
    caffe2::OperatorDef def;
    caffe2::AffineChannelNdOp<float, caffe2::CUDAContext> op(def, nullptr);
    caffe2::Tensor X((void*)x, {n, c, inner}), S((void*)scale, {scale_len}), B((void*)bias, {scale_len}), Y((void*)y, {});
    op.inputs_ = {&X, &S, &B};
    op.outputs_ = {&Y};
    op.RunOnDevice();
    return hipDeviceSynchronize() == hipSuccess ? 0 : -2;
  } catch (const std::exception& e) {
    if (err && err_len > 0) snprintf(err, err_len, "%s", e.what());
    return -1;
  }
}

int ref_affine_channel_nd_bwd(const float* scale, const float* dy, float* dx, int n, int c, int inner, int scale_len, char* err,
                              int err_len) {
  try {
    caffe2::OperatorDef def;
    caffe2::AffineChannelNdGradientOp<float, caffe2::CUDAContext> op(def, nullptr);
    caffe2::Tensor S((void*)scale, {scale_len}), DY((void*)dy, {n, c, inner}), DX((void*)dx, {});
    op.inputs_ = {&S, &DY};
    op.outputs_ = {&DX};
    op.RunOnDevice();
    return hipDeviceSynchronize() == hipSuccess ? 0 : -2;
  } catch (const std::exception& e) {
    if (err && err_len > 0) snprintf(err, err_len, "%s", e.what());
    return -1;
  }
}

}  // extern "C"
