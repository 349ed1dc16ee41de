// Context management for libmi355_isdf.so: stream, rocBLAS handle, hipFFT plan cache, workspace.
#include "common.h"
#include <cstdarg>

int isdf_fail(isdf_handle h, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) h->err = buf;
  else fprintf(stderr, "mi355_isdf: %s\n", buf);
  return code;
}

void* isdf_ws(isdf_handle h, const char* name, size_t bytes) {
  auto& slot = h->ws[name];
  if (slot.second >= bytes && slot.first) return slot.first;
  if (slot.first) {
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(slot.first);
    slot.first = nullptr;
    slot.second = 0;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
  if (e != hipSuccess) {
    isdf_fail(h, ISDF_ERR_HIP, "workspace '%s': hipMalloc(%zu) failed: %s", name, bytes,
              hipGetErrorString(e));
    return nullptr;
  }
  slot.first = p;
  slot.second = bytes;
  return p;
}

int isdf_get_plan(isdf_handle h, const int32_t mesh[3], int batch, FftPlan** out) {
  std::vector<int> key = {mesh[0], mesh[1], mesh[2], batch};
  auto it = h->plans.find(key);
  if (it != h->plans.end()) {
    *out = &it->second;
    return ISDF_OK;
  }
  FftPlan p;
  int dims[3] = {mesh[0], mesh[1], mesh[2]};
  int64_t G = (int64_t)mesh[0] * mesh[1] * mesh[2];
  int64_t Gc = (int64_t)mesh[0] * mesh[1] * (mesh[2] / 2 + 1);
  ARG_CHECK(h, G < (int64_t)2147483647);
  FFT_TRY(h, hipfftPlanMany(&p.fwd, 3, dims, nullptr, 1, (int)G, nullptr, 1, (int)Gc, HIPFFT_D2Z, batch));
  hipfftResult r2 = hipfftPlanMany(&p.bwd, 3, dims, nullptr, 1, (int)Gc, nullptr, 1, (int)G, HIPFFT_Z2D, batch);
  if (r2 == HIPFFT_SUCCESS) r2 = hipfftSetStream(p.fwd, h->stream);
  if (r2 == HIPFFT_SUCCESS) r2 = hipfftSetStream(p.bwd, h->stream);
  if (r2 != HIPFFT_SUCCESS) {                       // do not leak the half-built pair
    (void)hipfftDestroy(p.fwd);
    if (p.bwd) (void)hipfftDestroy(p.bwd);
    return isdf_fail(h, ISDF_ERR_LIB, "hipFFT plan (%d x %d x %d, batch %d) failed: result %d", mesh[0], mesh[1], mesh[2], batch,
                     (int)r2);
  }
  auto res = h->plans.emplace(key, p);
  *out = &res.first->second;
  return ISDF_OK;
}

extern "C" {

int isdf_abi_version(void) { return 19; }

int isdf_set_coulomb_omega(isdf_handle h, double omega) {
  if (!h) return ISDF_ERR_ARG;
  h->coul_omega = omega;
  return ISDF_OK;
}

int isdf_set_coulomb_cutoff(isdf_handle h, double rc) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, rc >= 0.0);
  h->coul_rc = rc;
  return ISDF_OK;
}

int isdf_set_coulomb_ws(isdf_handle h, double alpha, const double ak[9], const int32_t mesh[3], const double maxq[3],
                        const double* d_vq) {
  if (!h) return ISDF_ERR_ARG;
  if (alpha <= 0.0) { h->wsk = WsKernel(); return ISDF_OK; }
  ARG_CHECK(h, ak && mesh && maxq && d_vq && mesh[0] > 0 && mesh[1] > 0 && mesh[2] > 0);
  h->wsk.alpha = alpha;
  for (int i = 0; i < 9; ++i) h->wsk.ak[i] = ak[i];
  for (int i = 0; i < 3; ++i) { h->wsk.mesh[i] = mesh[i]; h->wsk.maxq[i] = maxq[i]; }
  h->wsk.vq = d_vq;
  return ISDF_OK;
}

int isdf_set_option(isdf_handle h, const char* key, int value) {
  if (!h) return ISDF_ERR_ARG;
  ARG_CHECK(h, key != nullptr);
  if (std::string(key) == "trsm_substitution") { h->trsm_substitution = value ? 1 : 0; return ISDF_OK; }
  if (std::string(key) == "own_fft") { h->own_fft = value < 0 ? 0 : (value > 2 ? 2 : value); return ISDF_OK; }
  if (std::string(key) == "gemm_nn_own") { h->gemm_nn_own = value ? 1 : 0; return ISDF_OK; }
  if (std::string(key) == "coul_sphere") { h->coul_sphere = value < 0 ? 0 : value; return ISDF_OK; }
  if (std::string(key) == "conv_pipe") { h->conv_pipe = value != 0; return ISDF_OK; }
  if (std::string(key) == "conv_sub_rows") { h->conv_sub_rows = value < 0 ? 0 : value; return ISDF_OK; }
  if (std::string(key) == "gram_pivot_tpb") {
    if (value != 64 && value != 128 && value != 256) return ISDF_ERR_ARG;
    h->gram_pivot_tpb = value;
    return ISDF_OK;
  }
  if (std::string(key) == "block_apply_waves") { h->block_apply_waves = value < 1 ? 1 : value; return ISDF_OK; }
  if (std::string(key) == "block_apply_reg") { h->block_apply_reg = value ? 1 : 0; return ISDF_OK; }
  return isdf_fail(h, ISDF_ERR_ARG, "isdf_set_option: unknown key '%s'", key);
}

int isdf_create(int device_id, isdf_handle* out) {
  if (!out) return ISDF_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return isdf_fail(nullptr, ISDF_ERR_HIP, "no HIP device visible (%s)", hipGetErrorString(e));
  if (device_id < 0 || device_id >= ndev)
    return isdf_fail(nullptr, ISDF_ERR_ARG, "device %d out of range (%d devices)", device_id, ndev);
  isdf_ctx* h = new isdf_ctx();
  h->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess) { delete h; return ISDF_ERR_HIP; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) h->num_cu = prop.multiProcessorCount;
  if (rocblas_create_handle(&h->blas) != rocblas_status_success) {
    delete h;
    return isdf_fail(nullptr, ISDF_ERR_LIB, "rocblas_create_handle failed");
  }
  rocblas_set_pointer_mode(h->blas, rocblas_pointer_mode_host);
  *out = h;
  return ISDF_OK;
}

int isdf_release_workspace(isdf_handle h) {
  if (!h) return ISDF_ERR_ARG;
  (void)hipStreamSynchronize(h->stream);
  for (auto& kv : h->ws)
    if (kv.second.first) (void)hipFree(kv.second.first);
  h->ws.clear();
  for (auto& kv : h->plans) {
    if (kv.second.fwd) hipfftDestroy(kv.second.fwd);
    if (kv.second.bwd) hipfftDestroy(kv.second.bwd);
  }
  h->plans.clear();
  return ISDF_OK;
}

int isdf_destroy(isdf_handle h) {
  if (!h) return ISDF_OK;
  isdf_prof_reset(h);
  isdf_release_workspace(h);
  if (h->blas) rocblas_destroy_handle(h->blas);
  delete h;
  return ISDF_OK;
}

int isdf_set_stream(isdf_handle h, void* hip_stream) {
  if (!h) return ISDF_ERR_ARG;
  h->stream = (hipStream_t)hip_stream;
  BLAS_TRY(h, rocblas_set_stream(h->blas, h->stream));
  for (auto& kv : h->plans) {
    if (kv.second.fwd) FFT_TRY(h, hipfftSetStream(kv.second.fwd, h->stream));
    if (kv.second.bwd) FFT_TRY(h, hipfftSetStream(kv.second.bwd, h->stream));
  }
  return ISDF_OK;
}

const char* isdf_last_error(isdf_handle h) { return h ? h->err.c_str() : "null handle"; }

int isdf_prof_enable(isdf_handle h, int on) {
  if (!h) return ISDF_ERR_ARG;
  h->profiling = on ? 1 : 0;
  return ISDF_OK;
}

int isdf_prof_reset(isdf_handle h) {
  if (!h) return ISDF_ERR_ARG;
  (void)hipStreamSynchronize(h->stream);
  for (auto& kv : h->prof)
    for (auto& r : kv.second.recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  h->prof.clear();
  return ISDF_OK;
}

int isdf_prof_count(isdf_handle h) { return h ? (int)h->prof.size() : 0; }

int isdf_prof_get(isdf_handle h, int index, char* name, int name_cap, int64_t* launches, double* total_ms,
                  double* total_work) {
  if (!h || index < 0 || index >= (int)h->prof.size()) return ISDF_ERR_ARG;
  auto it = h->prof.begin();
  std::advance(it, index);
  (void)hipStreamSynchronize(h->stream);
  double ms = 0.0;
  for (auto& r : it->second.recs) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) ms += t;
  }
  if (name && name_cap > 0) { strncpy(name, it->first.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (launches) *launches = it->second.launches;
  if (total_ms) *total_ms = ms;
  if (total_work) *total_work = it->second.work;
  return ISDF_OK;
}

int64_t isdf_workspace_bytes(isdf_handle h) {
  if (!h) return 0;
  int64_t s = 0;
  for (auto& kv : h->ws) s += (int64_t)kv.second.second;
  return s;
}

}  // extern "C"
