// asdr_front_host.cpp -- host side + C ABI (include/asdr_front.h) of the batched AudioSDRpreProcessor,
// AudioIQgenerator and AudioGrabberComplex256.  Control-plane members live in a host
// mirror that is pushed before a launch when a setter touched it and pulled back lazily when a getter needs what
// the kernels changed.  There is no CPU implementation of the signal paths: update calls need a HIP device.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "asdr_front_device.h"

int asdr_internal_fail(const std::string &m);   // asdr_host.cpp: sets the thread's asdr_last_error() text, returns -1

namespace {
int fail(const std::string &m) { return asdr_internal_fail(m); }
#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));              \
  } while (0)

int ensure_tables() {   // __constant__ tables are per device: (re)uploaded for the current device at every create (a few hundred bytes)
  if (asdr_front_upload_tables() != 0) return fail("front-end table upload failed");
  return 0;
}

struct DevBase {
  int n = 0, device = ASDR_NO_DEVICE;
  hipStream_t stream = nullptr, last_stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool ev_valid = false;
  int16_t *d_io[4] = {nullptr, nullptr, nullptr, nullptr};   // staging for the host-pointer entry points
  size_t io_cap = 0;
};

int dev_init(DevBase &d, int n, int device) {
  d.n = n; d.device = device;
  if (device == ASDR_NO_DEVICE) return 0;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail("no HIP device: this library has no CPU fallback");
  if (device < 0 || device >= count) return fail("bad device index");
  HIPCHK(hipSetDevice(device));
  if (ensure_tables() != 0) return -1;
  HIPCHK(hipStreamCreate(&d.stream));
  HIPCHK(hipEventCreate(&d.ev0));
  HIPCHK(hipEventCreate(&d.ev1));
  return 0;
}
void dev_fini(DevBase &d) {
  if (d.device == ASDR_NO_DEVICE) return;
  hipSetDevice(d.device);
  hipDeviceSynchronize();
  for (int16_t *p : d.d_io) if (p) hipFree(p);
  if (d.ev0) hipEventDestroy(d.ev0);
  if (d.ev1) hipEventDestroy(d.ev1);
  if (d.stream) hipStreamDestroy(d.stream);
}
int dev_stage(DevBase &d, size_t count) {   // staging buffers of `count` int16 each
  if (count <= d.io_cap) return 0;
  HIPCHK(hipStreamSynchronize(d.stream));
  for (int i = 0; i < 4; i++) {
    if (d.d_io[i]) HIPCHK(hipFree(d.d_io[i]));
    d.d_io[i] = nullptr;
    HIPCHK(hipMalloc(&d.d_io[i], count * sizeof(int16_t)));
  }
  d.io_cap = count;
  return 0;
}
int dev_sync(DevBase &d) {
  if (d.device == ASDR_NO_DEVICE) return 0;
  HIPCHK(hipSetDevice(d.device));
  HIPCHK(hipStreamSynchronize(d.last_stream));   // nullptr = the null stream
  HIPCHK(hipStreamSynchronize(d.stream));
  return 0;
}
float dev_last_ms(DevBase &d) {
  if (!d.ev_valid) return -1.0f;
  float ms = -1.0f;
  if (hipEventSynchronize(d.ev1) != hipSuccess) return -1.0f;
  if (hipEventElapsedTime(&ms, d.ev0, d.ev1) != hipSuccess) return -1.0f;
  return ms;
}
int check_io(const char *what, int n_blocks, long in_stride, long out_stride, uintptr_t ptr_bits) {
  if (in_stride < n_blocks || out_stride < n_blocks) return fail(std::string(what) + ": row stride shorter than n_blocks");
  if (in_stride > 0x7fffffffL || out_stride > 0x7fffffffL) return fail(std::string(what) + ": row stride too large");
  if ((ptr_bits & 15u) != 0) return fail(std::string(what) + ": device pointers must be 16-byte aligned");
  return 0;
}
const char *kNoDevice = "control-plane-only batch (ASDR_NO_DEVICE): the signal path needs a HIP device";
}  // namespace

// ============================== AudioSDRpreProcessor ==============================
struct asdr_pre_batch : DevBase {
  std::vector<asdr_pre_state_t> host;   // mirror of the device state
  asdr_pre_state_t *d_state = nullptr;
  bool host_dirty = true;               // a setter changed the mirror since the last push
  bool device_newer = false;            // a launch may have changed the device state since the last pull
};

namespace {
int pre_pull(asdr_pre_batch *p) {
  if (p->device == ASDR_NO_DEVICE || !p->device_newer) return 0;
  if (dev_sync(*p) != 0) return -1;
  HIPCHK(hipMemcpy(p->host.data(), p->d_state, p->n * sizeof(asdr_pre_state_t), hipMemcpyDeviceToHost));
  p->device_newer = false;
  return 0;
}
template <typename F>
void pre_each(asdr_pre_batch *p, int ch, F f) {
  if (!p) return;
  if (ch != ASDR_ALL && (ch < 0 || ch >= p->n)) return;
  if (pre_pull(p) != 0) return;
  if (ch == ASDR_ALL) for (int i = 0; i < p->n; i++) f(p->host[i]);
  else f(p->host[ch]);
  p->host_dirty = true;
}
}  // namespace

extern "C" {

asdr_pre_t *asdr_pre_create(int n_channels, int device) {
  if (n_channels <= 0 || n_channels > (1 << 20)) { fail("n_channels must be in 1..1048576"); return nullptr; }
  asdr_pre_batch *p = new asdr_pre_batch();
  if (dev_init(*p, n_channels, device) != 0) { delete p; return nullptr; }
  asdr_pre_state_t z;
  memset(&z, 0, sizeof z);              // AudioSDRpreProcessor.h:72-83: correction 0, counters 0, no swap, autoDetectFlag false
  p->host.assign(n_channels, z);
  if (device != ASDR_NO_DEVICE && hipMalloc(&p->d_state, n_channels * sizeof(asdr_pre_state_t)) != hipSuccess) {
    fail("out of device memory"); dev_fini(*p); delete p; return nullptr;
  }
  return p;
}

void asdr_pre_destroy(asdr_pre_t *p) {
  if (!p) return;
  dev_fini(*p);
  if (p->d_state) hipFree(p->d_state);
  delete p;
}

int asdr_pre_n_channels(const asdr_pre_t *p) { return p ? p->n : 0; }

int asdr_pre_update_device(asdr_pre_t *p, const int16_t *dI, const int16_t *dQ, int16_t *dIout, int16_t *dQout, int n_blocks,
                           long in_stride_blocks, long out_stride_blocks, void *stream_) {
  if (!p) return fail("null batch");
  if (p->device == ASDR_NO_DEVICE) return fail(kNoDevice);
  if (!dI || !dQ) return 0;             // AudioSDRpreProcessor.cpp:50-52
  if (!dIout || !dQout) return fail("null output");
  if (n_blocks <= 0) return 0;
  if (check_io("asdr_pre_update_device", n_blocks, in_stride_blocks, out_stride_blocks,
               (uintptr_t)dI | (uintptr_t)dQ | (uintptr_t)dIout | (uintptr_t)dQout) != 0) return -1;
  hipStream_t stream = (hipStream_t)stream_;
  HIPCHK(hipSetDevice(p->device));
  if (p->host_dirty) {
    if (pre_pull(p) != 0) return -1;    // (a setter always pulls first; this is for the very first launch)
    HIPCHK(hipMemcpyAsync(p->d_state, p->host.data(), p->n * sizeof(asdr_pre_state_t), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));   // the mirror may be modified again right after this call returns
    p->host_dirty = false;
  }
  PreArgs a;
  a.state = p->d_state; a.in_i = dI; a.in_q = dQ; a.out_i = dIout; a.out_q = dQout;
  a.n_channels = p->n; a.n_blocks = n_blocks; a.in_stride = (int32_t)in_stride_blocks; a.out_stride = (int32_t)out_stride_blocks;
  HIPCHK(hipEventRecord(p->ev0, stream));
  if (asdr_launch_pre(&a, stream) != 0) return fail("pre-processor kernel launch failed");
  HIPCHK(hipEventRecord(p->ev1, stream));
  p->ev_valid = true; p->last_stream = stream; p->device_newer = true;
  return 0;
}

int asdr_pre_update(asdr_pre_t *p, int16_t *I, int16_t *Q, int n_blocks) {
  if (!p) return fail("null batch");
  if (p->device == ASDR_NO_DEVICE) return fail(kNoDevice);
  if (!I || !Q) return 0;
  if (n_blocks <= 0) return 0;
  HIPCHK(hipSetDevice(p->device));
  const size_t count = (size_t)p->n * n_blocks * ASDR_N;
  if (dev_stage(*p, count) != 0) return -1;
  HIPCHK(hipMemcpyAsync(p->d_io[0], I, count * sizeof(int16_t), hipMemcpyHostToDevice, p->stream));
  HIPCHK(hipMemcpyAsync(p->d_io[1], Q, count * sizeof(int16_t), hipMemcpyHostToDevice, p->stream));
  if (asdr_pre_update_device(p, p->d_io[0], p->d_io[1], p->d_io[0], p->d_io[1], n_blocks, n_blocks, n_blocks, p->stream) != 0) return -1;
  HIPCHK(hipMemcpyAsync(I, p->d_io[0], count * sizeof(int16_t), hipMemcpyDeviceToHost, p->stream));
  HIPCHK(hipMemcpyAsync(Q, p->d_io[1], count * sizeof(int16_t), hipMemcpyDeviceToHost, p->stream));
  HIPCHK(hipStreamSynchronize(p->stream));
  return 0;
}

int asdr_pre_synchronize(asdr_pre_t *p) { return p ? dev_sync(*p) : fail("null batch"); }
float asdr_pre_last_kernel_ms(asdr_pre_t *p) { return p ? dev_last_ms(*p) : -1.0f; }

void asdr_pre_startAutoI2SerrorDetection(asdr_pre_t *p, int ch) {
  pre_each(p, ch, [](asdr_pre_state_t &s) { s.auto_detect = 1; s.correction = 0; s.failure_count = 0; s.success_count = 0; });
}
void asdr_pre_stopAutoI2SerrorDetection(asdr_pre_t *p, int ch) {
  pre_each(p, ch, [](asdr_pre_state_t &s) { s.auto_detect = 0; s.correction = 0; });
}
void asdr_pre_setI2SerrorCompensation(asdr_pre_t *p, int ch, int correction) {
  pre_each(p, ch, [&](asdr_pre_state_t &s) { s.correction = (int16_t)correction; s.auto_detect = 0; });
}
void asdr_pre_swapIQ(asdr_pre_t *p, int ch, int swap) {
  pre_each(p, ch, [&](asdr_pre_state_t &s) { s.swap = swap ? 1 : 0; });
}
int asdr_pre_getAutoI2SerrorDetectionStatus(asdr_pre_t *p, int ch) {
  if (!p || ch < 0 || ch >= p->n || pre_pull(p) != 0) return 0;
  return p->host[ch].auto_detect;
}
int16_t asdr_pre_getI2SerrorCompensation(asdr_pre_t *p, int ch) {
  if (!p || ch < 0 || ch >= p->n || pre_pull(p) != 0) return 0;
  return p->host[ch].correction;
}
int asdr_pre_read_state(asdr_pre_t *p, asdr_pre_state_t *dst) {
  if (!p) return fail("null batch");
  if (!dst) return fail("null destination");
  if (pre_pull(p) != 0) return -1;
  memcpy(dst, p->host.data(), p->n * sizeof(asdr_pre_state_t));
  return 0;
}

}  // extern "C"

// ============================== AudioIQgenerator ==============================
struct asdr_iqgen_batch : DevBase {
  std::vector<float> gains;   // [n][2]: gainI, gainQ (AudioIQgenerator.h:69-70)
  float *d_gains = nullptr;
  int16_t *d_hist = nullptr;   // [n][2][128]: raw ring of the two carried blocks
  uint32_t phase = 0;          // slot of the older one
  bool gains_dirty = true;
};

extern "C" {

asdr_iqgen_t *asdr_iqgen_create(int n_channels, int device) {
  if (n_channels <= 0 || n_channels > (1 << 20)) { fail("n_channels must be in 1..1048576"); return nullptr; }
  asdr_iqgen_batch *g = new asdr_iqgen_batch();
  if (dev_init(*g, n_channels, device) != 0) { delete g; return nullptr; }
  g->gains.assign(2 * (size_t)n_channels, 1.0f);
  if (device != ASDR_NO_DEVICE) {
    if (hipMalloc(&g->d_gains, 2 * (size_t)n_channels * sizeof(float)) != hipSuccess ||
        hipMalloc(&g->d_hist, 256 * (size_t)n_channels * sizeof(int16_t)) != hipSuccess ||
        hipMemset(g->d_hist, 0, 256 * (size_t)n_channels * sizeof(int16_t)) != hipSuccess) {   // static buffers start zeroed (.cpp:37-38)
      fail("out of device memory"); if (g->d_gains) hipFree(g->d_gains); if (g->d_hist) hipFree(g->d_hist); dev_fini(*g); delete g; return nullptr;
    }
    hipDeviceSynchronize();
  }
  return g;
}
void asdr_iqgen_destroy(asdr_iqgen_t *g) {
  if (!g) return;
  dev_fini(*g);
  if (g->d_gains) hipFree(g->d_gains);
  if (g->d_hist) hipFree(g->d_hist);
  delete g;
}
int asdr_iqgen_n_channels(const asdr_iqgen_t *g) { return g ? g->n : 0; }

void asdr_iqgen_setGainBalance(asdr_iqgen_t *g, int ch, float balance) {   // AudioIQgenerator.h:55-59
  if (!g) return;
  if (ch != ASDR_ALL && (ch < 0 || ch >= g->n)) return;
  const float gi = balance, gq = (float)(1.0 / (double)balance);
  for (int i = (ch == ASDR_ALL ? 0 : ch); i < (ch == ASDR_ALL ? g->n : ch + 1); i++) { g->gains[2 * i] = gi; g->gains[2 * i + 1] = gq; }
  g->gains_dirty = true;
}

int asdr_iqgen_update_device(asdr_iqgen_t *g, const int16_t *dIn, int16_t *dI, int16_t *dQ, int n_blocks, long in_stride_blocks,
                             long out_stride_blocks, void *stream_) {
  if (!g) return fail("null batch");
  if (g->device == ASDR_NO_DEVICE) return fail(kNoDevice);
  if (!dIn) return 0;                   // AudioIQgenerator.cpp:43-45
  if (!dI || !dQ) return fail("null output");
  if (n_blocks <= 0) return 0;
  if (check_io("asdr_iqgen_update_device", n_blocks, in_stride_blocks, out_stride_blocks, (uintptr_t)dIn | (uintptr_t)dI | (uintptr_t)dQ) != 0) return -1;
  hipStream_t stream = (hipStream_t)stream_;
  HIPCHK(hipSetDevice(g->device));
  if (g->gains_dirty) {
    HIPCHK(hipMemcpyAsync(g->d_gains, g->gains.data(), g->gains.size() * sizeof(float), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    g->gains_dirty = false;
  }
  IqgenArgs a;
  a.hist = g->d_hist; a.phase = g->phase; a.gains = g->d_gains; a.in = dIn; a.out_i = dI; a.out_q = dQ;
  a.n_channels = g->n; a.n_blocks = n_blocks; a.in_stride = (int32_t)in_stride_blocks; a.out_stride = (int32_t)out_stride_blocks;
  HIPCHK(hipEventRecord(g->ev0, stream));
  if (asdr_launch_iqgen(&a, stream) != 0) return fail("IQ generator kernel launch failed");
  g->phase = (g->phase + (uint32_t)n_blocks) & 1u;
  HIPCHK(hipEventRecord(g->ev1, stream));
  g->ev_valid = true; g->last_stream = stream;
  return 0;
}

int asdr_iqgen_update(asdr_iqgen_t *g, const int16_t *in, int16_t *I, int16_t *Q, int n_blocks) {
  if (!g) return fail("null batch");
  if (g->device == ASDR_NO_DEVICE) return fail(kNoDevice);
  if (!in) return 0;
  if (!I || !Q) return fail("null output");
  if (n_blocks <= 0) return 0;
  HIPCHK(hipSetDevice(g->device));
  const size_t count = (size_t)g->n * n_blocks * ASDR_N;
  if (dev_stage(*g, count) != 0) return -1;
  HIPCHK(hipMemcpyAsync(g->d_io[0], in, count * sizeof(int16_t), hipMemcpyHostToDevice, g->stream));
  if (asdr_iqgen_update_device(g, g->d_io[0], g->d_io[1], g->d_io[2], n_blocks, n_blocks, n_blocks, g->stream) != 0) return -1;
  HIPCHK(hipMemcpyAsync(I, g->d_io[1], count * sizeof(int16_t), hipMemcpyDeviceToHost, g->stream));
  HIPCHK(hipMemcpyAsync(Q, g->d_io[2], count * sizeof(int16_t), hipMemcpyDeviceToHost, g->stream));
  HIPCHK(hipStreamSynchronize(g->stream));
  return 0;
}
int asdr_iqgen_synchronize(asdr_iqgen_t *g) { return g ? dev_sync(*g) : fail("null batch"); }
float asdr_iqgen_last_kernel_ms(asdr_iqgen_t *g) { return g ? dev_last_ms(*g) : -1.0f; }

}  // extern "C"

// ============================== AudioGrabberComplex256 ==============================
struct asdr_grab_batch : DevBase {
  int16_t *d_buffer = nullptr, *d_out = nullptr;
  float *d_spec = nullptr;          // staging for the host-destination spectrum call
  int parity = 0;                   // _buffStart / 256: identical for every channel, updates are batch-wide
  bool valid = false;               // _dataBufferValid
  std::vector<uint8_t> new_data;    // _newDataIsAvailable per channel
};

extern "C" {

asdr_grab_t *asdr_grab_create(int n_channels, int device) {
  if (n_channels <= 0 || n_channels > (1 << 20)) { fail("n_channels must be in 1..1048576"); return nullptr; }
  asdr_grab_batch *g = new asdr_grab_batch();
  if (dev_init(*g, n_channels, device) != 0) { delete g; return nullptr; }
  g->new_data.assign(n_channels, 0);
  if (device != ASDR_NO_DEVICE) {
    const size_t bytes = 512 * (size_t)n_channels * sizeof(int16_t);
    if (hipMalloc(&g->d_buffer, bytes) != hipSuccess || hipMalloc(&g->d_out, bytes) != hipSuccess ||
        hipMemset(g->d_buffer, 0, bytes) != hipSuccess || hipMemset(g->d_out, 0, bytes) != hipSuccess) {
      fail("out of device memory"); if (g->d_buffer) hipFree(g->d_buffer); if (g->d_out) hipFree(g->d_out); dev_fini(*g); delete g; return nullptr;
    }
    hipDeviceSynchronize();
  }
  return g;
}
void asdr_grab_destroy(asdr_grab_t *g) {
  if (!g) return;
  dev_fini(*g);
  if (g->d_buffer) hipFree(g->d_buffer);
  if (g->d_out) hipFree(g->d_out);
  if (g->d_spec) hipFree(g->d_spec);
  delete g;
}
int asdr_grab_n_channels(const asdr_grab_t *g) { return g ? g->n : 0; }

int asdr_grab_update_device(asdr_grab_t *g, const int16_t *dI, const int16_t *dQ, int n_blocks, long in_stride_blocks, void *stream_) {
  if (!g) return fail("null batch");
  if (g->device == ASDR_NO_DEVICE) return fail(kNoDevice);
  if (!dI || !dQ) return 0;             // AudioGrabberComplex256.cpp:54-56
  if (n_blocks <= 0) return 0;
  if (check_io("asdr_grab_update_device", n_blocks, in_stride_blocks, in_stride_blocks, (uintptr_t)dI | (uintptr_t)dQ) != 0) return -1;
  hipStream_t stream = (hipStream_t)stream_;
  HIPCHK(hipSetDevice(g->device));
  GrabArgs a;
  a.buffer = g->d_buffer; a.out_buffer = g->d_out; a.in_i = dI; a.in_q = dQ;
  a.n_channels = g->n; a.n_blocks = n_blocks; a.in_stride = (int32_t)in_stride_blocks; a.parity = g->parity;
  HIPCHK(hipEventRecord(g->ev0, stream));
  if (asdr_launch_grab(&a, stream) != 0) return fail("grabber kernel launch failed");
  HIPCHK(hipEventRecord(g->ev1, stream));
  g->ev_valid = true; g->last_stream = stream;
  const int total = g->parity + n_blocks;
  if (total >= 2) { g->valid = true; std::fill(g->new_data.begin(), g->new_data.end(), (uint8_t)1); }   // .cpp:62-68
  g->parity = total & 1;
  return 0;
}

int asdr_grab_update(asdr_grab_t *g, const int16_t *I, const int16_t *Q, int n_blocks) {
  if (!g) return fail("null batch");
  if (g->device == ASDR_NO_DEVICE) return fail(kNoDevice);
  if (!I || !Q) return 0;
  if (n_blocks <= 0) return 0;
  HIPCHK(hipSetDevice(g->device));
  const size_t count = (size_t)g->n * n_blocks * ASDR_N;
  if (dev_stage(*g, count) != 0) return -1;
  HIPCHK(hipMemcpyAsync(g->d_io[0], I, count * sizeof(int16_t), hipMemcpyHostToDevice, g->stream));
  HIPCHK(hipMemcpyAsync(g->d_io[1], Q, count * sizeof(int16_t), hipMemcpyHostToDevice, g->stream));
  if (asdr_grab_update_device(g, g->d_io[0], g->d_io[1], n_blocks, n_blocks, g->stream) != 0) return -1;
  HIPCHK(hipStreamSynchronize(g->stream));
  return 0;
}

int asdr_grab_newDataAvailable(asdr_grab_t *g, int ch) { return (g && ch >= 0 && ch < g->n) ? g->new_data[ch] : 0; }

int asdr_grab_grab(asdr_grab_t *g, int ch, int16_t *destination) {
  if (!g) return fail("null batch");
  if (ch < 0 || ch >= g->n) return fail("bad channel");
  if (!destination) return fail("null destination");
  int copied = 0;
  if (g->valid) {                       // .cpp:81-86
    if (g->device == ASDR_NO_DEVICE) return fail(kNoDevice);
    if (dev_sync(*g) != 0) return -1;
    HIPCHK(hipMemcpy(destination, g->d_out + (size_t)ch * 512, 512 * sizeof(int16_t), hipMemcpyDeviceToHost));
    copied = 1;
  }
  g->new_data[ch] = 0;                  // .cpp:88
  return copied;
}

int asdr_grab_grab_all(asdr_grab_t *g, int16_t *destination) {
  if (!g) return fail("null batch");
  if (!destination) return fail("null destination");
  int copied = 0;
  if (g->valid) {
    if (g->device == ASDR_NO_DEVICE) return fail(kNoDevice);
    if (dev_sync(*g) != 0) return -1;
    HIPCHK(hipMemcpy(destination, g->d_out, (size_t)g->n * 512 * sizeof(int16_t), hipMemcpyDeviceToHost));
    copied = 1;
  }
  std::fill(g->new_data.begin(), g->new_data.end(), (uint8_t)0);
  return copied;
}

const int16_t *asdr_grab_device_ptr(asdr_grab_t *g) { return g ? g->d_out : nullptr; }

int asdr_grab_power_spectrum_device(asdr_grab_t *g, float *dDestination, void *stream_) {
  if (!g) return fail("null batch");
  if (g->device == ASDR_NO_DEVICE) return fail(kNoDevice);
  if (!dDestination) return fail("null destination");
  if (((uintptr_t)dDestination & 15u) != 0) return fail("asdr_grab_power_spectrum_device: device pointers must be 16-byte aligned");
  if (!g->valid) return 0;
  hipStream_t stream = (hipStream_t)stream_;
  HIPCHK(hipSetDevice(g->device));
  if (stream != g->last_stream) HIPCHK(hipStreamSynchronize(g->last_stream));   // the buffers were written on the update stream
  if (asdr_launch_grab_spectrum(g->d_out, dDestination, g->n, stream) != 0) return fail("spectrum kernel launch failed");
  return 1;
}

int asdr_grab_power_spectrum(asdr_grab_t *g, float *destination) {
  if (!g) return fail("null batch");
  if (g->device == ASDR_NO_DEVICE) return fail(kNoDevice);
  if (!destination) return fail("null destination");
  if (!g->valid) return 0;
  HIPCHK(hipSetDevice(g->device));
  const size_t bytes = (size_t)g->n * 256 * sizeof(float);
  if (!g->d_spec) HIPCHK(hipMalloc(&g->d_spec, bytes));
  if (asdr_grab_power_spectrum_device(g, g->d_spec, g->stream) != 1) return -1;
  HIPCHK(hipMemcpyAsync(destination, g->d_spec, bytes, hipMemcpyDeviceToHost, g->stream));
  HIPCHK(hipStreamSynchronize(g->stream));
  return 1;
}
int asdr_grab_synchronize(asdr_grab_t *g) { return g ? dev_sync(*g) : fail("null batch"); }

}  // extern "C"
