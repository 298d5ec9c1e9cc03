// asdr_host.cpp -- host control plane + C ABI of libasdr_hip.so (see include/asdr.h).
//
// Mirrors the reference's `AudioSDR` class surface (SRC/AudioSDRlib/AudioSDR.h:88-156,
// AudioSDR.cpp:174-757) for a batch of N channels: every setter is a field write on a host-side
// per-channel record (plus the same derived-parameter arithmetic the reference performs, in the same
// float/double types), and marks the device parameter block dirty; update() uploads what changed and
// launches the HIP kernels of asdr_kernels.hip.  There is deliberately no CPU implementation of the
// signal path here: without a HIP device asdr_create() fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/asdr.h"
#include "asdr_device.h"
#include "asdr_tables.h"

// batches of at least this many waves issue a multi-block call as one launch per block (asdr_update_device_strided)
#ifndef ASDR_PER_BLOCK_LAUNCH_WAVES
#define ASDR_PER_BLOCK_LAUNCH_WAVES 1024
#endif
// A multi-block call on a SMALL batch of SSB-class channels runs as a block pipeline of three role-specialised waves per channel
// group (asdr_stream_kernel): all 3 x waves workgroups must be resident at once, hence the cap; shorter calls are not worth the
// pipeline's fill and drain.
#ifndef ASDR_STREAM_MAX_WAVES
#define ASDR_STREAM_MAX_WAVES 1024   /* upper bound on what the occupancy query may allow (asdr_create) */
#endif
#ifndef ASDR_STREAM_MIN_BLOCKS
#define ASDR_STREAM_MIN_BLOCKS 8
#endif
extern "C" int asdr_launch_stream(const UpdateArgs *a, hipStream_t stream);
extern "C" int asdr_stream_capacity(int device, int *compute_units);
extern "C" int asdr_launch_stream_snapshot(const UpdateArgs *a, void *snap, int restore, hipStream_t stream);
extern "C" int asdr_launch_stream_ack(uint32_t *err, hipStream_t stream);
extern "C" int asdr_kernels_upload_tables(void);
extern "C" int asdr_launch_update(const UpdateArgs *a, int variant, int uniform, hipStream_t stream);
extern "C" int asdr_launch_reset(const UpdateArgs *a, const uint32_t *d_reset_bits, int first_row, int n_rows, hipStream_t stream);

namespace {

constexpr double kPI = 3.1415926535897932384626433832795;  // Arduino.h PI
constexpr float kFs = 44100.0f;                            // AUDIO_SAMPLE_RATE_EXACT (Teensy 4.x)
constexpr float kIFcenter = 6890.0f, kBWssb = 3000.0f, kBWcw = 1000.0f, kBWwspr = 1000.0f, kBWam = 8500.0f;  // .h:164-168

thread_local std::string g_err;
int fail(const std::string &m) { g_err = m; return -1; }
#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));              \
  } while (0)

// One reference instance's control-plane members (AudioSDR.h:172-246), batch-side naming.
struct Chan {
  float in_gain = 1.0f, in_gain_i = 1.0f, in_gain_q = 1.0f, gain_balance = 1.0f;
  float output_gain = 0.5f, out_gain = 1.0f, current_out_gain = 1.0f;
  float freq_shift = 0.0f;
  uint16_t mode = 0;
  bool muted = true;
  int16_t current_filter = 0;
  bool af_en = false;
  int if_table = ASDR_TBL_IF_SSB, audio_table = ASDR_TBL_AUDIO_BASE + ASDR_audio2700;
  int16_t als_m = 55, als_delay = 3;
  float als_lambda = 0.5f;
  bool als_en = false, als_notch = true, als_adaptive = true;
  float agc_alpha_att = 0, agc_beta_att = 0, agc_alpha_rel = 0, agc_beta_rel = 0;
  float agc_attack_ms = 0, agc_release_ms = 0, agc_knee = 0, agc_slope = 0, agc_threshold = 0;
  float agc_static_gain = 10.0f;
  float agc_slot129 = 100.0f;  // _agc_hangTime, which is also _agc_gainLookup[129] (.h:219-220)
  uint32_t agc_hang_count = 0;
  bool agc_en = true;
  int agc_table = -1;
  float nb_threshold = 1.2f;
  bool nb_en = true;
};

struct AgcTable {
  float thr, slope, knee;
  float v[ASDR_AGC_TAB_ROW];
};

// log2_approx_f32, AudioSDR.h:483-491
float log2_approx(float input) {
  int exponent;
  float mantissa = frexpf(fabsf(input), &exponent);
  return (((1.23149591368684f * mantissa - 4.11852516267426f) * mantissa + 6.02197014179219f) * mantissa -
          3.13396450166353f) + exponent;
}

// agc_createLookupTable, AudioSDR.cpp:459-480 (130 entries: the loop bound is tableSize + 1)
void build_agc_table(AgcTable &t) {
  const float thr = t.thr, slope = t.slope, knee = t.knee;
  const float lin_lo = expf((float)(2.3025 * ((double)thr - (double)knee / 2.0) / 20.0));
  const float lin_hi = expf((float)(2.3025 * ((double)thr + (double)knee / 2.0) / 20.0));
  for (int i = 0; i < 130; i++) {
    const float input = (float)((double)(float)i / 128.0);
    const float in_db = (float)(6.026 * (double)log2_approx(input));
    if (input < lin_lo) {
      t.v[i] = 1.0f;
    } else if (input > lin_hi) {
      const float out_db = thr + (in_db - thr) * slope;
      t.v[i] = expf((float)(2.3025 * (double)(out_db - in_db) / 20.0));
    } else {
      const double u = (double)(in_db - thr) + (double)knee / 2.0;
      const float out_db = (float)((double)in_db + (((double)slope - 1.0) * u * u) / (2.0 * (double)knee));
      t.v[i] = expf((float)(2.3025 * (double)(out_db - in_db) / 20.0));
    }
  }
  t.v[130] = t.v[131] = 0.0f;
}

float time_constant(float ms) {  // AudioSDR.cpp:448/553: exp(log(0.1) / (FS*ms/1000.0))
  return (float)exp(log(0.1) / ((double)(kFs * ms) / 1000.0));
}

}  // namespace

#define ASDR_AUX_STREAMS (ASDR_KERNEL_KINDS + 1)   /* every sub-range but the first runs on a helper stream */
struct asdr_batch {
  int n = 0, device = 0;
  hipStream_t stream = nullptr;  // used by the host-pointer entry point and by getters
  std::vector<Chan> ch;          // n + 1 (last = dummy channel used to pad the last wave)
  std::vector<AgcTable> agc_pool;
  std::unordered_map<uint64_t, std::vector<int>> agc_index;   // hash of (threshold, slope, knee) bits -> rows of agc_pool
  std::vector<uint32_t> agc_refs;                              // channels using each table
  std::vector<ChanParams> hp;
  std::vector<uint32_t> reset;
  std::vector<SlotInfo> sched;
  // Incremental control plane: a setter marks only the channels it touched.  flush() refills and uploads those rows; the
  // wave schedule is rebuilt only when a touched channel's schedule key (kernel kind, mode, enables, tables) changed.
  std::vector<int32_t> dirty;          // channels whose parameter row changed since the last flush (each listed once)
  std::vector<uint8_t> dirty_flag;
  bool all_dirty = true;               // first flush, or so many dirty rows that one bulk upload is cheaper
  bool sched_dirty = true, reset_pending = true, agc_pool_dirty = true;
  bool agc_refs_changed = true;        // a setter moved a channel to another gain table since the last compaction check
  int reset_lo = 0x7fffffff, reset_hi = -1;   // rows with pending reset bits lie in [reset_lo, reset_hi]
  // The sorted schedule is launched as up to three sub-ranges, one per kernel instantiation (plain / SAM / ALS), each padded
  // to whole waves with the dummy channel: one SAM or ALS channel no longer demotes the whole batch.
  // Inside a sub-range the whole waves of each key group come first ("uniform" waves: 8 real channels, one key -> the
  // instantiation with scalar mode/flag tests), then the groups' remainders packed together ("mixed").
  int kind_first[ASDR_KERNEL_KINDS] = {}, kind_slots[ASDR_KERNEL_KINDS] = {}, kind_uniform_slots[ASDR_KERNEL_KINDS] = {};
  int left_first = 0, left_slots = 0, left_kind = ASDR_KERNEL_PLAIN;   // the key groups' remainders of all kinds: one sub-range, one launch
  bool kind_direct[ASDR_KERNEL_KINDS] = {};   // the uniform part is ONE key group of consecutive channel ids (checked when the schedule is built)
  // counters for the control-plane tests (ASDR_NO_DEVICE): what the last flush did
  long stat_rows_refilled = 0, stat_sched_rebuilds = 0, stat_bulk_uploads = 0;
  // device
  ChanParams *d_params = nullptr;
  ChanSmall *d_small = nullptr;
  int16_t *d_nb_hist = nullptr;
  uint8_t *d_nb_mask = nullptr;
  float *d_hil_q = nullptr, *d_hil_i = nullptr, *d_als_x = nullptr,
        *d_als_w = nullptr, *d_agc_tab = nullptr, *d_taps = nullptr;
  float *d_audio_prev = nullptr;   // _audioOut of the previous block, per channel (unknown mode values re-process it: AudioSDR.cpp:149-161)
  bool exact_unknown_mode = true;  // asdr_set_exact_unknown_mode
  size_t agc_tab_cap = 0;
  SlotInfo *d_sched = nullptr;
  uint32_t *d_reset = nullptr;
  int16_t *d_io[3] = {nullptr, nullptr, nullptr};
  int16_t *d_capture = nullptr;  // capture sink [n][capture_cap][128]
  long capture_cap = 0, capture_pos = 0;
  size_t io_cap = 0;
  bool taps_on = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool ev_valid = false;
  bool time_calls = false;       // asdr_set_launch_timing: an event pair around every call's launches (two more packets per call)
  hipEvent_t rev0 = nullptr, rev1 = nullptr;   // asdr_region_timing_begin / _end
  hipStream_t region_stream = nullptr;
  long region_calls = -1;        // update calls since asdr_region_timing_begin (-1: no region open)
  std::vector<hipEvent_t> tev;   // optional per-launch event pairs (asdr_kernel_timing_begin)
  size_t tev_used = 0;
  hipStream_t last_stream = nullptr;
  hipEvent_t ev_last = nullptr;  // recorded after every launch: a call on another stream waits for it first
  hipStream_t aux[ASDR_AUX_STREAMS] = {};   // helper streams for concurrent sub-range launches
  hipEvent_t ev_fork = nullptr, ev_join[ASDR_AUX_STREAMS] = {};
  bool ev_last_valid = false;
  LoEntry *d_lo = nullptr;       // local-oscillator cache, 2 entries (asdr_device.h)
  // streaming pipeline (asdr_stream_kernel): exchange rings and progress counters, allocated at its first use
  float *d_xch_a = nullptr, *d_xch_b = nullptr;
  uint32_t *d_stream_prog = nullptr;
  LoEntry *d_lo_ring = nullptr;
  // Launch-form switches.  Their defaults come from the environment WHEN THE BATCH IS CREATED (ASDR_SAM_FUSED, ASDR_SAM_SPLIT_MIN,
  // ASDR_NO_STREAM_PIPELINE: comparison switches of the measurement tools); asdr_set_sam_launch_form / asdr_set_stream_pipeline change
  // them per batch afterwards.
  bool sam_fused = false;           // SAM channels through the fused 4-wave kernel instead of the pre | PLL | post launches
  int sam_split_min = 512;          // ... which are chosen from this many SAM channels on
  bool stream_pipeline = true;      // small multi-block calls may run as the block pipeline (asdr_stream_kernel)
  int stream_max_waves = 0;         // channel groups the pipeline may hold: 3 w + 1 workgroups must be co-resident, one per compute unit
  uint32_t stream_spin_limit = ASDR_STREAM_SPIN_LIMIT;
  void *d_stream_snap = nullptr;    // snapshot of the state a pipeline call advances (asdr_kernels.hip "the pipeline as a transaction")
  long stat_stream_recoveries = 0;  // pipeline calls that gave up and were re-run on the in-kernel block loop (read back at synchronisation points)
  bool sam_split = false;           // decided when the schedule is built (enough SAM channels, not sam_fused)
  int als_split_min = 0x7fffffff;   // channels with a short ALS filter (not SAM) run as chain | filter launches from this many on (default: never --
                                    // measured: 2 % faster for C4's 131,072-channel share, 6 % slower at 1,048,576 channels, 3 % slower all-ALS)
  bool als_split = false;           // decided when the schedule is built
  float *d_xch_sam = nullptr;       // SAM sub-range as pre | PLL | post launches: the IF rows of the current block (1 KB per slot)
  size_t xch_sam_slots = 0;
  bool plain_uniform_ssb = false;   // every uniform wave of the plain instantiation runs an SSB-class mode or AM: the modes the block pipeline has roles for (checked when the schedule is built)
  bool stream_launched = false;     // a streaming launch is (or was) in flight: its error flag has not been read yet
  long stat_stream_launches = 0;
  uint32_t lo_parity = 0;
  uint32_t nb_phase = 0;         // blocks processed so far, mod 3 (position of every channel's blanker ring)
  uint32_t als_phase = 0;        // blocks processed so far, mod 2 (position of every channel's ALS input ring)
  ChainConsts k{};
};

namespace {

uint64_t agc_hash(float thr, float slope, float knee) {
  uint32_t a, b2, c;
  memcpy(&a, &thr, 4); memcpy(&b2, &slope, 4); memcpy(&c, &knee, 4);
  uint64_t h = 1469598103934665603ull;
  for (uint32_t w : {a, b2, c}) { h ^= w; h *= 1099511628211ull; }
  return h;
}

int find_agc_table(asdr_batch *b, float thr, float slope, float knee) {
  std::vector<int> &bucket = b->agc_index[agc_hash(thr, slope, knee)];
  for (int i : bucket) {
    const AgcTable &t = b->agc_pool[i];
    if (memcmp(&t.thr, &thr, 4) == 0 && memcmp(&t.slope, &slope, 4) == 0 && memcmp(&t.knee, &knee, 4) == 0) return i;
  }
  AgcTable t;
  t.thr = thr; t.slope = slope; t.knee = knee;
  build_agc_table(t);
  b->agc_pool.push_back(t);
  b->agc_refs.push_back(0);
  bucket.push_back((int)b->agc_pool.size() - 1);
  b->agc_pool_dirty = true;
  return (int)b->agc_pool.size() - 1;
}

void rebuild_agc(asdr_batch *b, Chan &c) {  // agc_createLookupTable(): also clobbers _agc_hangTime's storage
  const int old = c.agc_table;
  c.agc_table = find_agc_table(b, c.agc_threshold, c.agc_slope, c.agc_knee);
  b->agc_refs[c.agc_table]++;
  b->agc_refs_changed = true;
  if (old >= 0 && old < (int)b->agc_refs.size() && b->agc_refs[old] > 0) b->agc_refs[old]--;
  c.agc_slot129 = b->agc_pool[c.agc_table].v[129];
}

// Drop tables no channel uses any more (a UI knob sweep leaves one dead table per step) once they outnumber the live ones.
void compact_agc_pool(asdr_batch *b) {
  if (!b->agc_refs_changed) return;   // (every update call comes through here: no scan of the pool unless a reference moved)
  b->agc_refs_changed = false;
  size_t live = 0;
  for (uint32_t r : b->agc_refs) live += (r != 0);
  if (b->agc_pool.size() < 64 || b->agc_pool.size() <= 2 * live) return;
  std::vector<int> remap(b->agc_pool.size(), -1);
  std::vector<AgcTable> pool;
  std::vector<uint32_t> refs;
  for (size_t i = 0; i < b->agc_pool.size(); i++)
    if (b->agc_refs[i]) { remap[i] = (int)pool.size(); pool.push_back(b->agc_pool[i]); refs.push_back(b->agc_refs[i]); }
  b->agc_pool.swap(pool); b->agc_refs.swap(refs);
  b->agc_index.clear();
  for (size_t i = 0; i < b->agc_pool.size(); i++) b->agc_index[agc_hash(b->agc_pool[i].thr, b->agc_pool[i].slope, b->agc_pool[i].knee)].push_back((int)i);
  for (Chan &c : b->ch) c.agc_table = remap[c.agc_table];
  b->agc_pool_dirty = true; b->all_dirty = true; b->sched_dirty = true;   // every row carries a table index
}

void mark_reset(asdr_batch *b, int i, uint32_t bits) {
  b->reset[i] |= bits;
  b->reset_pending = true;
  if (i < b->reset_lo) b->reset_lo = i;
  if (i > b->reset_hi) b->reset_hi = i;
}

// setDemodMode, AudioSDR.cpp:187-222
float set_mode(asdr_batch *b, int idx, int new_mode) {
  Chan &c = b->ch[idx];
  c.mode = (uint16_t)new_mode;
  int tbl = -1;
  if (c.mode == ASDR_USBmode) { c.freq_shift = (float)((double)kIFcenter - (double)kBWssb / 2.0); tbl = ASDR_TBL_IF_SSB; }
  else if (c.mode == ASDR_LSBmode) { c.freq_shift = (float)((double)kIFcenter + (double)kBWssb / 2.0); tbl = ASDR_TBL_IF_SSB; }
  else if (c.mode == ASDR_WSPRmode) { c.freq_shift = (float)((double)kIFcenter - (double)kBWssb / 2.0); tbl = ASDR_TBL_IF_WSPR; }
  else if (c.mode == ASDR_CW_USBmode) { c.freq_shift = (float)((double)kIFcenter - (double)kBWcw / 2.0); tbl = ASDR_TBL_IF_CW; }
  else if (c.mode == ASDR_CW_LSBmode) { c.freq_shift = (float)((double)kIFcenter + (double)kBWcw / 2.0); tbl = ASDR_TBL_IF_CW; }
  else if (c.mode == ASDR_AMmode || c.mode == ASDR_SAMmode) { c.freq_shift = kIFcenter; tbl = ASDR_TBL_IF_AM; }
  if (tbl >= 0) { c.if_table = tbl; mark_reset(b, idx, ASDR_R_IF); }
  return c.freq_shift;
}

// agc_init, AudioSDR.cpp:439-457
void agc_init(asdr_batch *b, Chan &c) {
  c.agc_threshold = -60.0f; c.agc_slope = 0.1f; c.agc_knee = 2.0f;
  c.agc_attack_ms = 5.0f; c.agc_release_ms = 500.0f;
  c.agc_slot129 = 100.0f;
  c.agc_hang_count = (uint32_t)((double)kFs * ((double)c.agc_slot129 / 1000.0));
  c.agc_alpha_att = time_constant(c.agc_attack_ms);
  c.agc_beta_att = (float)(1.0 - (double)c.agc_alpha_att);
  c.agc_alpha_rel = time_constant(c.agc_release_ms);
  c.agc_beta_rel = (float)(1.0 - (double)c.agc_alpha_rel);
  c.agc_en = true;
  rebuild_agc(b, c);
}

// init(), AudioSDR.cpp:174-185
void chan_init(asdr_batch *b, int idx) {
  Chan &c = b->ch[idx];
  c.audio_table = ASDR_TBL_AUDIO_BASE + ASDR_audio2700;
  c.if_table = ASDR_TBL_IF_SSB;
  mark_reset(b, idx, ASDR_R_AF | ASDR_R_IF | ASDR_R_IMG | ASDR_R_NB);
  agc_init(b, c);
  set_mode(b, idx, ASDR_LSBmode);
  c.muted = false;
}

void fill_params(const Chan &c, ChanParams &p) {
  memset(&p, 0, sizeof p);
  p.mode = c.mode;
  p.flags = (c.nb_en ? ASDR_F_NB_EN : 0u) | (c.af_en ? ASDR_F_AF_EN : 0u) | (c.agc_en ? ASDR_F_AGC_EN : 0u) |
            (c.als_en ? ASDR_F_ALS_EN : 0u) | (c.als_notch ? ASDR_F_ALS_NOTCH : 0u) |
            (c.als_adaptive ? ASDR_F_ALS_ADAPTIVE : 0u) | (c.muted ? ASDR_F_MUTED : 0u);
  p.if_table = c.if_table; p.audio_table = c.audio_table; p.agc_table = c.agc_table;
  p.in_gain_i = c.in_gain_i; p.in_gain_q = c.in_gain_q; p.output_gain = c.output_gain;
  p.freq_shift = c.freq_shift; p.nb_threshold = c.nb_threshold;
  p.agc_alpha_att = c.agc_alpha_att; p.agc_beta_att = c.agc_beta_att;
  p.agc_alpha_rel = c.agc_alpha_rel; p.agc_beta_rel = c.agc_beta_rel;
  p.agc_static_gain = c.agc_static_gain; p.agc_hang_count = c.agc_hang_count;
  p.als_m = c.als_m; p.als_delay = c.als_delay; p.als_lambda = c.als_lambda;
}

// Wave scheduling: channels are grouped by kernel instantiation (plain / SAM / ALS) and, inside it, by (mode, enables, tables),
// so that a wave's 8 channels take the same branches.  Rows of I/Q/out and of every state array are per channel, so the
// grouping costs nothing in coalescing.
int kernel_kind(const ChanParams &p) {
  if (p.flags & ASDR_F_ALS_EN) {
    // a short filter on a channel that needs no PLL fits the plain instantiation's LDS rows (12 instead of 9 waves per CU)
    const bool small = p.als_m >= 0 && p.als_m <= 64 && p.als_delay >= 0 && p.als_delay + p.als_m <= 65;
    return !small ? ASDR_KERNEL_ALS : (p.mode == ASDR_SAMmode ? ASDR_KERNEL_SAM_ALS : ASDR_KERNEL_ALS_SMALL);
  }
  return (p.mode == ASDR_SAMmode) ? ASDR_KERNEL_SAM : ASDR_KERNEL_PLAIN;
}
#ifndef ASDR_SAM_SPLIT_MIN_CHANNELS
#define ASDR_SAM_SPLIT_MIN_CHANNELS 512
#endif
uint64_t sched_key(const ChanParams &p) {
  return ((uint64_t)kernel_kind(p) << 60) | ((uint64_t)(p.mode & 0xFFFF) << 40) | ((uint64_t)(p.flags & 0xFF) << 32) |
         ((uint64_t)(p.if_table & 0xFF) << 24) | ((uint64_t)(p.audio_table & 0xFF) << 16) | (uint64_t)(p.agc_table & 0xFFFF);
}

void mark_dirty(asdr_batch *b, int i) {
  if (b->all_dirty) return;
  if (!b->dirty_flag[i]) { b->dirty_flag[i] = 1; b->dirty.push_back(i); }
  if (b->dirty.size() > (size_t)b->n / 8 + 64) { b->all_dirty = true; }   // a bulk refill is cheaper from here on
}

// Host half of flush(): refill the parameter rows that changed, decide whether the schedule has to be rebuilt, rebuild it.
// Returns the sorted list of refilled rows in `rows_out` (empty + all_dirty = every row).  No HIP call in here, so the
// control-plane tests can time it on a device-less batch.
void flush_host(asdr_batch *b, std::vector<int32_t> &rows_out, bool &bulk, bool &sched_rebuilt) {
  const int rows = b->n + 1;
  compact_agc_pool(b);
  bulk = b->all_dirty;
  rows_out.clear();
  b->stat_rows_refilled = 0;
  if (bulk) {
    for (int i = 0; i < rows; i++) {
      const uint64_t old_key = sched_key(b->hp[i]);
      fill_params(b->ch[i], b->hp[i]);
      if (i < b->n && sched_key(b->hp[i]) != old_key) b->sched_dirty = true;
    }
    b->stat_rows_refilled = rows;
    b->stat_bulk_uploads++;
  } else if (!b->dirty.empty()) {
    std::sort(b->dirty.begin(), b->dirty.end());
    for (int32_t i : b->dirty) {
      const uint64_t old_key = sched_key(b->hp[i]);
      fill_params(b->ch[i], b->hp[i]);
      if (sched_key(b->hp[i]) != old_key) b->sched_dirty = true;
      b->dirty_flag[i] = 0;
    }
    rows_out = b->dirty;
    b->stat_rows_refilled = (long)rows_out.size();
  }
  b->dirty.clear();
  if (bulk) std::fill(b->dirty_flag.begin(), b->dirty_flag.end(), 0);
  b->all_dirty = false;
  sched_rebuilt = false;
  if (b->sched_dirty) {
    std::vector<std::pair<uint64_t, int32_t>> order(b->n);
    for (int i = 0; i < b->n; i++) order[i] = {sched_key(b->hp[i]), i};
    std::sort(order.begin(), order.end());   // (key, channel): equal keys stay in channel order
    // per kernel kind: slots of the whole waves of every key group, and of the groups' remainders
    int uni[ASDR_KERNEL_KINDS] = {}, rem[ASDR_KERNEL_KINDS] = {};
    // SAM (no ALS) as three launches per block only when there are enough SAM channels to fill the PLL kernel's waves: a handful
    // of them is quicker through the fused kernel (one launch, no exchange round trips)
    int n_sam = 0;
    for (int i = 0; i < b->n; i++) { const int k = (int)(order[i].first >> 60); n_sam += (k == ASDR_KERNEL_SAM || k == ASDR_KERNEL_SAM_ALS); }
    b->sam_split = !b->sam_fused && n_sam >= b->sam_split_min;
    {   // short ALS filters on channels that are not SAM: the chain up to the AGC as the plain instantiation, then the filter as a
        // launch of its own (asdr_als_kernel: small LDS rows, 18 waves per CU) -- when there are enough of them to fill it
      int n_als_small = 0;
      for (int i = 0; i < b->n; i++) n_als_small += ((int)(order[i].first >> 60) == ASDR_KERNEL_ALS_SMALL);
      b->als_split = n_als_small >= b->als_split_min;
    }
    const bool sam_general_only = !b->sam_split;
    // The remainders (< 8 channels) of all key groups share ONE sub-range behind the kinds' whole waves, run by one launch of a
    // general instantiation: as a launch per kind they were up to five more streams of a handful of long-lived waves each, and
    // streams that share a hardware queue run one after the other (profiles/README.md: C4's last 80 us were four such waves).
    int left = 0; bool left_general = false;
    for (int i = 0; i < b->n;) {
      int j = i + 1;
      while (j < b->n && order[j].first == order[i].first) j++;
      const int k = (int)(order[i].first >> 60), g = j - i;
      if (k == ASDR_KERNEL_SAM && sam_general_only) rem[k] += g;   // the fused SAM instantiation has only the general form (4-wave workgroups)
      else { uni[k] += g / 8 * 8; left += g % 8; if (g % 8) left_general = left_general || (k != ASDR_KERNEL_PLAIN); }
      i = j;
    }
    int pos = 0;
    for (int k = 0; k < ASDR_KERNEL_KINDS; k++) {
      b->kind_first[k] = pos; b->kind_uniform_slots[k] = uni[k]; b->kind_slots[k] = uni[k] + ((rem[k] + 7) / 8) * 8;
      pos += b->kind_slots[k];
    }
    b->left_first = pos; b->left_slots = (left + 7) / 8 * 8; b->left_kind = left_general ? ASDR_KERNEL_ALS : ASDR_KERNEL_PLAIN;
    pos += b->left_slots;
    b->sched.assign(pos, SlotInfo{b->n, b->hp[b->n].mode, b->hp[b->n].flags, 0u});   // padding = the dummy channel
    int at_u[ASDR_KERNEL_KINDS], at_m[ASDR_KERNEL_KINDS], at_left = b->left_first;
    for (int k = 0; k < ASDR_KERNEL_KINDS; k++) { at_u[k] = b->kind_first[k]; at_m[k] = b->kind_first[k] + uni[k]; }
    // Local-oscillator cache: the settings groups with whole waves get an entry each, largest first come first served (the mixer's
    // pairs are a function of (phase, increment); receivers of one group that were configured together share them for ever)
    uint32_t lo_next = 0;
    for (int i = 0; i < b->n;) {
      int j = i + 1;
      while (j < b->n && order[j].first == order[i].first) j++;
      const int k = (int)(order[i].first >> 60), g = j - i;
      const bool fused = (k == ASDR_KERNEL_SAM && sam_general_only);
      const int whole = fused ? 0 : g / 8 * 8;
      const uint32_t gm = b->hp[order[i].second].mode;
      const bool mixes_early = (gm == ASDR_USBmode || gm == ASDR_LSBmode || gm == ASDR_CW_USBmode || gm == ASDR_CW_LSBmode || gm == ASDR_WSPRmode || gm == ASDR_AMmode);
      const uint32_t lo_id = (whole > 0 && mixes_early && lo_next < ASDR_LO_ENTRIES) ? ++lo_next : 0u;   // 1 + entry, 0 = none
      for (int t = 0; t < g; t++) {
        const int c = order[i + t].second;
        int &at = (t < whole) ? at_u[k] : (fused ? at_m[k] : at_left);
        const uint32_t lo = (t < whole) ? (lo_id | ((t < 8 && lo_id) ? ASDR_LO_WRITER : 0u)) : 0u;
        b->sched[at++] = SlotInfo{c, b->hp[c].mode, b->hp[c].flags, lo};
      }
      i = j;
    }
    for (int k = 0; k < ASDR_KERNEL_KINDS; k++) {
      const SlotInfo *sl = b->sched.data() + b->kind_first[k];
      bool direct = b->kind_uniform_slots[k] > 0;
      for (int j = 1; j < b->kind_uniform_slots[k] && direct; j++)
        direct = (sl[j].ch == sl[0].ch + j) && sl[j].mode == sl[0].mode && sl[j].flags == sl[0].flags;
      b->kind_direct[k] = direct;
    }
    {
      const SlotInfo *sl = b->sched.data() + b->kind_first[ASDR_KERNEL_PLAIN];
      bool ssb = b->kind_uniform_slots[ASDR_KERNEL_PLAIN] > 0;
      for (int j = 0; j < b->kind_uniform_slots[ASDR_KERNEL_PLAIN] && ssb; j += 8) {
        const uint32_t m = sl[j].mode;
        ssb = (m == ASDR_USBmode || m == ASDR_LSBmode || m == ASDR_CW_USBmode || m == ASDR_CW_LSBmode || m == ASDR_WSPRmode ||
               m == ASDR_AMmode);   // (round 3: the pipeline's role 2 also runs the envelope detector)
      }
      b->plain_uniform_ssb = ssb;
    }
    b->sched_dirty = false;
    sched_rebuilt = true;
    b->stat_sched_rebuilds++;
  }
}

int flush(asdr_batch *b, hipStream_t stream) {
  const int rows = b->n + 1;
  std::vector<int32_t> changed;
  bool bulk = false, sched_rebuilt = false;
  flush_host(b, changed, bulk, sched_rebuilt);
  bool uploaded = false;
  if (b->agc_pool_dirty) {
    const size_t need = b->agc_pool.size() * ASDR_AGC_TAB_ROW;
    if (need > b->agc_tab_cap) {
      HIPCHK(hipStreamSynchronize(stream));      // ordered behind every earlier launch (update waits for ev_last first)
      if (b->d_agc_tab) HIPCHK(hipFree(b->d_agc_tab));
      b->agc_tab_cap = need * 2;
      HIPCHK(hipMalloc(&b->d_agc_tab, b->agc_tab_cap * sizeof(float)));
    }
    std::vector<float> flat(need);
    for (size_t i = 0; i < b->agc_pool.size(); i++) memcpy(&flat[i * ASDR_AGC_TAB_ROW], b->agc_pool[i].v, sizeof(float) * ASDR_AGC_TAB_ROW);
    HIPCHK(hipMemcpyAsync(b->d_agc_tab, flat.data(), need * sizeof(float), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));  // `flat` is a temporary
    b->agc_pool_dirty = false;
  }
  if (bulk) {
    HIPCHK(hipMemcpyAsync(b->d_params, b->hp.data(), rows * sizeof(ChanParams), hipMemcpyHostToDevice, stream));
    uploaded = true;
  } else {
    for (size_t i = 0; i < changed.size();) {   // one copy per run of consecutive rows
      size_t j = i + 1;
      while (j < changed.size() && changed[j] == changed[j - 1] + 1) j++;
      HIPCHK(hipMemcpyAsync(b->d_params + changed[i], &b->hp[changed[i]], (j - i) * sizeof(ChanParams), hipMemcpyHostToDevice, stream));
      uploaded = true;
      i = j;
    }
  }
  if (sched_rebuilt) {
    HIPCHK(hipMemcpyAsync(b->d_sched, b->sched.data(), b->sched.size() * sizeof(SlotInfo), hipMemcpyHostToDevice, stream));
    uploaded = true;
  }
  if (uploaded) HIPCHK(hipStreamSynchronize(stream));  // the host rows may be rewritten by the next setter + flush
  return 0;
}

void fill_args(asdr_batch *b, UpdateArgs &a) {
  memset(&a, 0, sizeof a);
  a.params = b->d_params; a.small = b->d_small;
  a.nb_hist = b->d_nb_hist; a.nb_mask = b->d_nb_mask; a.hil_q = b->d_hil_q; a.hil_i = b->d_hil_i;
  a.als_x = b->d_als_x; a.als_w = b->d_als_w; a.agc_tab = b->d_agc_tab;
  a.audio_prev = b->exact_unknown_mode ? b->d_audio_prev : nullptr;
  a.sched = b->d_sched; a.n_sched = 0; a.n_channels = b->n;   // the launcher sets the sub-range
  a.taps = b->taps_on ? b->d_taps : nullptr;
  a.nb_phase = b->nb_phase; a.als_phase = b->als_phase;
  a.lo_cache = b->d_lo; a.lo_parity = b->lo_parity; a.lo_write = 0;
  a.k = b->k;
}

int apply_resets(asdr_batch *b, hipStream_t stream) {
  if (!b->reset_pending) return 0;
  const int lo = b->reset_lo < 0 ? 0 : b->reset_lo, hi = b->reset_hi > b->n ? b->n : b->reset_hi;
  if (hi >= lo) {   // only the rows that carry bits travel and get a workgroup
    const int cnt = hi - lo + 1;
    HIPCHK(hipMemcpyAsync(b->d_reset + lo, b->reset.data() + lo, cnt * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    UpdateArgs a;
    fill_args(b, a);
    if (asdr_launch_reset(&a, b->d_reset, lo, cnt, stream) != 0) return fail("reset kernel launch failed");
    HIPCHK(hipStreamSynchronize(stream));
    std::fill(b->reset.begin() + lo, b->reset.begin() + hi + 1, 0u);
  }
  b->reset_pending = false; b->reset_lo = 0x7fffffff; b->reset_hi = -1;
  return 0;
}

template <typename F>
void each(asdr_batch *b, int ch, F f) {
  if (!b) return;
  if (ch == ASDR_ALL) { for (int i = 0; i < b->n; i++) f(i, b->ch[i]); b->all_dirty = true; }
  else if (ch >= 0 && ch < b->n) { f(ch, b->ch[ch]); mark_dirty(b, ch); }
  else return;
}
const Chan *get(asdr_batch *b, int ch) { return (b && ch >= 0 && ch < b->n) ? &b->ch[ch] : nullptr; }

// The streaming pipeline's recovery counter (a wave gave up waiting for its neighbour role, the call was re-run on the in-kernel
// block loop from the snapshot: asdr_kernels.hip "the pipeline as a transaction"): read back at the host's synchronisation points
// (asdr_synchronize, asdr_update, the status / capture readers).  Results are exact either way; this is only a statistic.
int check_stream_error(asdr_batch *b) {
  if (!b->stream_launched || !b->d_stream_prog) return 0;
  HIPCHK(hipStreamSynchronize(b->last_stream));
  uint32_t *word = b->d_stream_prog + 3 * ((b->n + 7) / 8) + 1, v[2] = {0, 0};
  HIPCHK(hipMemcpy(v, word, sizeof v, hipMemcpyDeviceToHost));
  b->stream_launched = false;
  b->stat_stream_recoveries = (long)v[1];
  if (v[0]) return fail("streaming pipeline: error word still set after the recovery launches (internal error)");
  return 0;
}

int read_small(asdr_batch *b, int ch, ChanSmall &s) {
  if (!b || ch < 0 || ch >= b->n) return fail("bad channel");
  if (b->device == ASDR_NO_DEVICE) return fail("control-plane-only batch has no device state");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->last_stream));   // nullptr = the null stream
  HIPCHK(hipStreamSynchronize(b->stream));
  if (check_stream_error(b) != 0) return -1;
  if (apply_resets(b, b->stream) != 0) return -1;
  HIPCHK(hipMemcpy(&s, b->d_small + ch, sizeof s, hipMemcpyDeviceToHost));
  return 0;
}

}  // namespace

extern "C" {

const char *asdr_last_error(void) { return g_err.c_str(); }
}  // extern "C"
// shared with asdr_front_host.cpp (internal C++ linkage, not part of the C ABI)
int asdr_internal_fail(const std::string &m) { return fail(m); }
extern "C" {
const char *asdr_version(void) { return "asdr-hip 0.1 (gfx950, wave64, -ffp-contract=off)"; }

asdr_batch_t *asdr_create(int n_channels, int device) {
  if (n_channels <= 0 || n_channels > (1 << 20)) { fail("n_channels must be in 1..1048576"); return nullptr; }
  asdr_batch *b = new asdr_batch();
  b->n = n_channels; b->device = device;
  b->sam_fused = getenv("ASDR_SAM_FUSED") != nullptr;
  b->sam_split_min = getenv("ASDR_SAM_SPLIT_MIN") ? atoi(getenv("ASDR_SAM_SPLIT_MIN")) : ASDR_SAM_SPLIT_MIN_CHANNELS;
  b->stream_pipeline = getenv("ASDR_NO_STREAM_PIPELINE") == nullptr;
  b->als_split_min = getenv("ASDR_ALS_SPLIT_MIN") ? atoi(getenv("ASDR_ALS_SPLIT_MIN")) : 0x7fffffff;
  const size_t rows = (size_t)n_channels + 1;
  if (device != ASDR_NO_DEVICE) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { delete b; fail("no HIP device: libasdr_hip has no CPU path"); return nullptr; }
    if (device < 0 || device >= ndev) { delete b; fail("bad device ordinal"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { delete b; fail("hipSetDevice failed"); return nullptr; }
    bool ok = true;
    auto alloc = [&](void **p, size_t bytes) { if (ok && hipMalloc(p, bytes) != hipSuccess) ok = false; };
    alloc((void **)&b->d_params, rows * sizeof(ChanParams));
    alloc((void **)&b->d_small, rows * sizeof(ChanSmall));
    alloc((void **)&b->d_nb_hist, rows * 768 * sizeof(int16_t));
    alloc((void **)&b->d_nb_mask, rows * ASDR_NB_MASK_ROW * sizeof(uint8_t));
    alloc((void **)&b->d_hil_q, rows * 256 * sizeof(float));
    alloc((void **)&b->d_hil_i, rows * 256 * sizeof(float));
    alloc((void **)&b->d_als_x, rows * 256 * sizeof(float));
    alloc((void **)&b->d_als_w, rows * 128 * sizeof(float));
    alloc((void **)&b->d_audio_prev, rows * 128 * sizeof(float));
    if (ok && hipMemset(b->d_audio_prev, 0, rows * 128 * sizeof(float)) != hipSuccess) ok = false;
    alloc((void **)&b->d_sched, (rows + 32) * sizeof(SlotInfo));   // three sub-ranges, each padded to a whole wave
    alloc((void **)&b->d_reset, rows * sizeof(uint32_t));
    alloc((void **)&b->d_lo, 2 * ASDR_LO_ENTRIES * sizeof(LoEntry));
    if (ok && hipMemset(b->d_lo, 0xFF, 2 * ASDR_LO_ENTRIES * sizeof(LoEntry)) != hipSuccess) ok = false;   // keys no phase can match
    if (ok && hipStreamCreate(&b->stream) != hipSuccess) ok = false;
    if (ok && hipEventCreate(&b->ev0) != hipSuccess) ok = false;
    if (ok && hipEventCreate(&b->ev1) != hipSuccess) ok = false;
    if (ok && hipEventCreate(&b->rev0) != hipSuccess) ok = false;
    if (ok && hipEventCreate(&b->rev1) != hipSuccess) ok = false;
    if (ok && hipEventCreateWithFlags(&b->ev_last, hipEventDisableTiming) != hipSuccess) ok = false;
    if (ok && hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming) != hipSuccess) ok = false;
    for (int i = 0; i < ASDR_AUX_STREAMS && ok; i++) {
      if (hipStreamCreateWithFlags(&b->aux[i], hipStreamNonBlocking) != hipSuccess) ok = false;
      if (ok && hipEventCreateWithFlags(&b->ev_join[i], hipEventDisableTiming) != hipSuccess) ok = false;
    }
    if (ok && asdr_kernels_upload_tables() != 0) ok = false;
    if (ok) {
      // The pipeline's roles wait for each other: its 3 w workgroups must fit the device at once -- an occupancy query, not an
      // assumption about the part (MI355X: 6 workgroups of this kernel per compute unit x 256 = 1,536 -> 512 channel groups).
      // Measured with several workgroups per compute unit (tools/stream_sizes.py): 6.9 us per block up to 84 groups, 7.3 at 128,
      // 8.4 at 256, 11.0 at 512 -- against 18.9-20.2 us for the in-kernel block loop at every one of those sizes.
      int cus = 0;
      const int cap = asdr_stream_capacity(device, &cus);
      b->stream_max_waves = cap >= 3 ? cap / 3 : 0;
      if (b->stream_max_waves > ASDR_STREAM_MAX_WAVES) b->stream_max_waves = ASDR_STREAM_MAX_WAVES;
    }
    if (!ok) { fail("device allocation / table upload failed"); asdr_destroy(b); return nullptr; }
  }

  // constants shared by all channels: in-class initialisers of AudioSDR.h:238-239, 249-284
  ChainConsts &k = b->k;
  k.if_center = kIFcenter;
  k.two_pi_f = (float)(2.0 * kPI);
  k.half_pi_f = (float)(0.5 * kPI);
  k.inv_two_pi_d = 1.0 / (double)k.two_pi_f;
  k.sin_index_scale_d = (65535.0 / (double)k.two_pi_f) * (1.0 + 0x1p-49);
  k.half_pi_d = kPI / 2.0;
  k.phase_inc_unit = k.two_pi_f / kFs;
  k.nb_alpha = 0.995f;
  k.nb_beta = (float)(1.0 - (double)k.nb_alpha);
  {
    const float wn = 0.07f, zeta = 0.707f, Ka = 1000.f;
    const float tau1 = Ka / (wn * wn);
    const float tau2 = 2 * zeta / wn;
    k.pll_b0 = (float)((double)(2 * Ka / tau1) * (1.0 + 2.0 * (double)tau2));
    k.pll_b1 = (float)((double)(2 * Ka / tau1) * (1.0 - 2.0 * (double)tau2));
    k.pll_a1 = -1.0f;
  }
  k.pll_alpha_freq = 0.995f;
  k.pll_beta_freq = (float)(1.0 - (double)k.pll_alpha_freq);
  k.pll_f_conv = kFs / k.two_pi_f;
  k.pll_lock_lo = (float)((double)kIFcenter - 1000.0);
  k.pll_lock_hi = (float)((double)kIFcenter + 1000.0);

  b->ch.assign(rows, Chan());
  b->hp.resize(rows);
  memset(b->hp.data(), 0, rows * sizeof(ChanParams));
  b->dirty_flag.assign(rows, 0);
  b->reset.assign(rows, ASDR_R_ALL);
  b->reset_lo = 0; b->reset_hi = (int)rows - 1;
  for (size_t i = 0; i < rows; i++) chan_init(b, (int)i);  // constructor -> init()
  b->all_dirty = true; b->sched_dirty = true; b->reset_pending = true;
  return b;
}

void asdr_destroy(asdr_batch_t *b) {
  if (!b) return;
  if (b->device == ASDR_NO_DEVICE) { delete b; return; }
  hipSetDevice(b->device);
  hipDeviceSynchronize();
  void *ptrs[] = {b->d_params, b->d_small, b->d_nb_hist, b->d_nb_mask, b->d_hil_q, b->d_hil_i, b->d_als_x, b->d_als_w,
                  b->d_agc_tab, b->d_taps, b->d_sched, b->d_reset, b->d_lo, b->d_io[0], b->d_io[1], b->d_io[2], b->d_capture,
                  b->d_xch_a, b->d_xch_b, b->d_stream_prog, b->d_lo_ring, b->d_xch_sam, b->d_audio_prev, b->d_stream_snap};
  for (void *p : ptrs) if (p) hipFree(p);
  for (hipEvent_t e : b->tev) hipEventDestroy(e);
  if (b->ev0) hipEventDestroy(b->ev0);
  if (b->ev1) hipEventDestroy(b->ev1);
  if (b->rev0) hipEventDestroy(b->rev0);
  if (b->rev1) hipEventDestroy(b->rev1);
  if (b->ev_last) hipEventDestroy(b->ev_last);
  if (b->ev_fork) hipEventDestroy(b->ev_fork);
  for (int i = 0; i < ASDR_AUX_STREAMS; i++) { if (b->ev_join[i]) hipEventDestroy(b->ev_join[i]); if (b->aux[i]) hipStreamDestroy(b->aux[i]); }
  if (b->stream) hipStreamDestroy(b->stream);
  delete b;
}

int asdr_n_channels(const asdr_batch_t *b) { return b ? b->n : 0; }

int asdr_update_device_strided(asdr_batch_t *b, const int16_t *dI, const int16_t *dQ, int16_t *dOut, int n_blocks,
                               long in_stride_blocks, long out_stride_blocks, void *stream_) {
  if (!b) return fail("null batch");
  if (b->device == ASDR_NO_DEVICE) return fail("control-plane-only batch (ASDR_NO_DEVICE): the signal path needs a HIP device");
  if (!dI || !dQ) return 0;  // missing-input guard, AudioSDR.cpp:48-56
  if (!dOut) return fail("null output");
  if (n_blocks <= 0) return 0;
  if (in_stride_blocks < n_blocks || out_stride_blocks < n_blocks) return fail("row stride shorter than n_blocks");
  if (in_stride_blocks > 0x7fffffffL || out_stride_blocks > 0x7fffffffL) return fail("row stride too large");
  if ((((uintptr_t)dI | (uintptr_t)dQ | (uintptr_t)dOut) & 15u) != 0) return fail("I/Q/out device pointers must be 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  HIPCHK(hipSetDevice(b->device));
  // one stream at a time per batch: a call on another stream first waits for the previous call's kernels (state in HBM is
  // read-modify-written by every launch)
  // (The event is recorded now, on the previous call's stream -- behind everything that call enqueued there -- rather than after
  // every call: one packet less per launch for the common single-stream caller.)
  if (b->ev_last_valid && stream != b->last_stream) {
    HIPCHK(hipEventRecord(b->ev_last, b->last_stream));
    HIPCHK(hipStreamWaitEvent(stream, b->ev_last, 0));
  }
  if (flush(b, stream) != 0) return -1;
  if (apply_resets(b, stream) != 0) return -1;
  UpdateArgs a;
  fill_args(b, a);
  a.in_i = dI; a.in_q = dQ; a.out = dOut; a.n_blocks = n_blocks;
  a.in_stride = (int32_t)in_stride_blocks; a.out_stride = (int32_t)out_stride_blocks;
  // Timing markers are opt-in: every event record is a packet the GPU's command processor handles between two kernels
  // (tools/launch_gap.py: 0.134 ms per back-to-back C2 call with a pair per call, 0.127 ms without).
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (b->tev_used + 2 <= b->tev.size()) { e0 = b->tev[b->tev_used]; e1 = b->tev[b->tev_used + 1]; b->tev_used += 2; }
  else if (b->time_calls) { e0 = b->ev0; e1 = b->ev1; }
  // Up to seven sub-ranges of the sorted schedule (the whole waves of 5 kernel kinds, the fused SAM kernel's general waves, the remainders).  They touch disjoint channels,
  // so they run CONCURRENTLY: the first on the caller's stream, the others on the batch's helper streams, forked behind an
  // event and joined before the call's end marker -- launched back to back on one stream the short ones (a handful of waves of
  // the slowest instantiation) would each add a whole wave lifetime to the step.  Heaviest instantiation first.
  struct Sub { int kind, uniform, first, slots; };
  Sub subs[ASDR_KERNEL_KINDS + 2]; int n_sub = 0;
  static const int heaviest_first[ASDR_KERNEL_KINDS] = {ASDR_KERNEL_ALS, ASDR_KERNEL_SAM_ALS, ASDR_KERNEL_ALS_SMALL, ASDR_KERNEL_SAM, ASDR_KERNEL_PLAIN};
  if (b->left_slots > 0) subs[n_sub++] = Sub{b->left_kind, 0, b->left_first, b->left_slots};   // a few long-lived waves: started first
  for (int kk = 0; kk < ASDR_KERNEL_KINDS; kk++) {
    const int k = heaviest_first[kk];
    if (b->kind_slots[k] == 0) continue;
    const int nu = b->kind_uniform_slots[k], nm = b->kind_slots[k] - nu;
    if (nu > 0) subs[n_sub++] = Sub{k, 1, b->kind_first[k], nu};
    if (nm > 0) subs[n_sub++] = Sub{k, 0, b->kind_first[k] + nu, nm};
  }
  // A multi-block call on a LARGE batch is issued as one launch per block: only the first block of a launch can read the
  // local-oscillator cache (wave 0 leaves the next LAUNCH's pairs), launches of >= 1024 waves follow each other without a gap
  // (tools/launch_gap.py), and C2 x 64 blocks ran at 0.157 ms per block inside one launch against 0.129 ms as 64 launches.  Small
  // batches (C5: 64 waves, 646 blocks) keep the in-kernel block loop: there a launch per 19-us block would be all overhead.
  int total_slots = 0;
  for (int i = 0; i < n_sub; i++) total_slots += subs[i].slots;
  // SAM channels (no ALS) run as three launches per block -- everything in front of the PLL | the PLL with one LANE per channel |
  // everything behind it -- through a 1 KB-per-slot exchange buffer: as a phase of one fused kernel the PLL's 128-step dependent
  // chain kept a workgroup's other waves waiting (ASDR_SAM_FUSED=1 selects that kernel, for comparison).
  const int sam_slots = b->kind_slots[ASDR_KERNEL_SAM] + b->kind_slots[ASDR_KERNEL_SAM_ALS];
  const bool sam_split = b->sam_split && sam_slots > 0;
  if (sam_split && (size_t)sam_slots > b->xch_sam_slots) {
    HIPCHK(hipStreamSynchronize(stream));
    float *grown = nullptr;
    HIPCHK(hipMalloc(&grown, (size_t)sam_slots * 2 * ASDR_N * sizeof(float)));   // (pointer and size change only once this succeeded)
    if (b->d_xch_sam) hipFree(b->d_xch_sam);
    b->d_xch_sam = grown; b->xch_sam_slots = (size_t)sam_slots;
  }
  a.xch_sam = sam_split ? b->d_xch_sam : nullptr;
  // ... and channels with a short ALS filter as two: the chain up to the AGC | the filter and the output stage (stage taps off: the taps
  // of the last two stages are the fused kernel's)
  const bool als_split = b->als_split && !b->taps_on && b->kind_uniform_slots[ASDR_KERNEL_ALS_SMALL] > 0;
  const bool per_block = n_blocks > 1 && (sam_split || als_split || total_slots >= 8 * ASDR_PER_BLOCK_LAUNCH_WAVES);
  // Small batch, many blocks, one sub-range of uniform SSB-class waves, no taps: the block pipeline -- as a transaction: a snapshot of
  // the state in front of it, and behind it the launches that put the state back and run the call on the in-kernel block loop if a
  // role's bounded wait fired (asdr_kernels.hip).  3 w + 1 workgroups must be co-resident: w <= stream_max_waves (occupancy query at
  // asdr_create, at most one workgroup per compute unit).
  if (b->stream_pipeline && n_blocks >= ASDR_STREAM_MIN_BLOCKS && n_sub == 1 && subs[0].kind == ASDR_KERNEL_PLAIN && subs[0].uniform &&
      b->plain_uniform_ssb &&
#ifndef ASDR_TIMELINE   /* (the profiling build writes its timestamps through the taps buffer: tools/timeline.py stream) */
      !b->taps_on &&
#endif
      subs[0].slots / 8 <= b->stream_max_waves) {
    const int w = subs[0].slots / 8;
    if (!b->d_xch_a) {   // first use: every buffer, or none (a failed allocation leaves the batch on the other launch forms)
      float *xa = nullptr, *xb = nullptr; uint32_t *prog = nullptr; LoEntry *ring = nullptr; void *snap = nullptr;
      const size_t n_prog = (size_t)(3 * ((b->n + 7) / 8) + 3);   // [.. + 1] = the error word, [.. + 2] = the recovery counter
      bool ok = hipMalloc(&xa, (size_t)(b->n + 1) * ASDR_STREAM_DEPTH * 2 * ASDR_N * sizeof(float)) == hipSuccess &&
                hipMalloc(&xb, (size_t)(b->n + 1) * ASDR_STREAM_DEPTH * (ASDR_N + 1) * sizeof(float)) == hipSuccess &&   // + the AM carrier words (xch_c) behind the rows
                hipMalloc(&prog, n_prog * sizeof(uint32_t)) == hipSuccess &&
                hipMalloc(&ring, ASDR_LO_RING * sizeof(LoEntry)) == hipSuccess &&
                hipMalloc(&snap, (size_t)(((b->n + 7) / 8) * 8) * ASDR_SNAP_BYTES) == hipSuccess &&
                hipMemsetAsync(prog, 0, n_prog * sizeof(uint32_t), stream) == hipSuccess;
      if (!ok) {
        for (void *p : {(void *)xa, (void *)xb, (void *)prog, (void *)ring, snap}) if (p) hipFree(p);
        return fail("streaming pipeline: out of device memory for its exchange rings");
      }
      b->d_xch_a = xa; b->d_xch_b = xb; b->d_stream_prog = prog; b->d_lo_ring = ring; b->d_stream_snap = snap;
    }
    HIPCHK(hipMemsetAsync(b->d_stream_prog, 0, (size_t)(3 * w + 1) * sizeof(uint32_t), stream));   // stream-ordered behind the previous launch
    a.sched = b->d_sched + subs[0].first; a.n_sched = subs[0].slots;
    a.direct_ch0 = -1;
    if (b->kind_direct[ASDR_KERNEL_PLAIN]) { const SlotInfo &s0 = b->sched[subs[0].first]; a.direct_ch0 = s0.ch; a.direct_mode = s0.mode; a.direct_flags = s0.flags; a.direct_lo = s0.lo; }
    a.lo_write = 0u;
    a.xch_a = b->d_xch_a; a.xch_b = b->d_xch_b; a.xch_c = b->d_xch_b + (size_t)(b->n + 1) * ASDR_STREAM_DEPTH * ASDR_N; a.stream_prog = b->d_stream_prog; a.stream_err = b->d_stream_prog + 3 * ((b->n + 7) / 8) + 1;
    a.stream_waves = w; a.lo_ring = b->d_lo_ring; a.stream_spin_limit = b->stream_spin_limit;
    if (e0) HIPCHK(hipEventRecord(e0, stream));
    if (asdr_launch_stream_snapshot(&a, b->d_stream_snap, 0, stream) != 0) return fail("stream snapshot launch failed");
    if (asdr_launch_stream(&a, stream) != 0) return fail("stream kernel launch failed");
    if (asdr_launch_stream_snapshot(&a, b->d_stream_snap, 1, stream) != 0) return fail("stream restore launch failed");
    {   // the same call on the in-kernel block loop, gated on the error word: its waves return at once when the pipeline completed
      UpdateArgs f = a;
      f.run_if = a.stream_err; f.stream_waves = 0; f.xch_a = nullptr; f.xch_b = nullptr; f.xch_c = nullptr; f.stream_prog = nullptr; f.lo_ring = nullptr;
      if (asdr_launch_update(&f, ASDR_KERNEL_PLAIN, 1, stream) != 0) return fail("stream fallback launch failed");
    }
    if (asdr_launch_stream_ack(a.stream_err, stream) != 0) return fail("stream acknowledge launch failed");
    if (e1) HIPCHK(hipEventRecord(e1, stream));
    b->stream_launched = true; b->stat_stream_launches++;
    b->ev_last_valid = true;
    b->ev_valid = (e0 != nullptr && e0 == b->ev0);
    if (b->region_calls >= 0) b->region_calls++;
    b->last_stream = stream;
    b->nb_phase = (b->nb_phase + (uint32_t)(n_blocks % 3)) % 3u;
    b->als_phase = (b->als_phase + (uint32_t)n_blocks) & 1u;
    b->lo_parity ^= 1u;
    return 0;
  }
  // The largest sub-range runs on the caller's stream: back-to-back calls then follow each other there without a gap, the helper
  // streams' joins (shorter kernels) are already satisfied when it ends, and only their starts pay the fork event's latency
  // (C4: 28 us per call with the remainders' launch on the caller's stream).
  int main_sub = 0;
  for (int i = 1; i < n_sub; i++) if (subs[i].slots > subs[main_sub].slots) main_sub = i;
  const int n_launch = per_block ? n_blocks : 1;
  float *const taps = a.taps;
  if (e0) HIPCHK(hipEventRecord(e0, stream));   // timing marker: right before the first launch
  for (int lb = 0; lb < n_launch; lb++) {
    if (per_block) {
      a.in_i = dI + (size_t)lb * ASDR_N; a.in_q = dQ + (size_t)lb * ASDR_N; a.out = dOut + (size_t)lb * ASDR_N; a.n_blocks = 1;
      a.nb_phase = (b->nb_phase + (uint32_t)(lb % 3)) % 3u;
      a.als_phase = (b->als_phase + (uint32_t)lb) & 1u;
      a.lo_parity = b->lo_parity ^ (uint32_t)(lb & 1);
      a.taps = (lb == n_launch - 1) ? taps : nullptr;   // the taps are those of the call's last block
    }
    if (n_sub > 1) HIPCHK(hipEventRecord(b->ev_fork, stream));
    for (int i = 0, n_aux = 0; i < n_sub; i++) {
      hipStream_t s = (i == main_sub) ? stream : b->aux[n_aux++];
      if (i != main_sub) HIPCHK(hipStreamWaitEvent(s, b->ev_fork, 0));
      a.sched = b->d_sched + subs[i].first; a.n_sched = subs[i].slots;
      a.direct_ch0 = -1;
      if (subs[i].uniform && b->kind_direct[subs[i].kind]) {   // one key group of consecutive channels: no schedule reads in the waves
        const SlotInfo &s0 = b->sched[subs[i].first];
        a.direct_ch0 = s0.ch; a.direct_mode = s0.mode; a.direct_flags = s0.flags; a.direct_lo = s0.lo;
      }
      a.lo_write = 1u;   // the first wave of every settings group fills the group's entry of the other half of the local-oscillator cache
      if (sam_split && subs[i].kind == ASDR_KERNEL_SAM)   // this sub-range's tiles (1 KB per slot, 8 slots per tile)
        a.xch_sam = b->d_xch_sam + (size_t)(subs[i].first - b->kind_first[ASDR_KERNEL_SAM]) * 2 * ASDR_N;
      if (sam_split && subs[i].kind == ASDR_KERNEL_SAM_ALS)   // ... behind those of the SAM kind
        a.xch_sam = b->d_xch_sam + (size_t)(b->kind_slots[ASDR_KERNEL_SAM] + subs[i].first - b->kind_first[ASDR_KERNEL_SAM_ALS]) * 2 * ASDR_N;
      const int form = (als_split && subs[i].kind == ASDR_KERNEL_ALS_SMALL && subs[i].uniform) ? 2 : subs[i].uniform;
      if (asdr_launch_update(&a, subs[i].kind, form, s) != 0) return fail("update kernel launch failed");
      if (i != main_sub) HIPCHK(hipEventRecord(b->ev_join[n_aux - 1], s));
    }
    for (int j = 0; j + 1 < n_sub; j++) HIPCHK(hipStreamWaitEvent(stream, b->ev_join[j], 0));   // behind the caller's stream's own launch
  }
  if (e1) HIPCHK(hipEventRecord(e1, stream));
  b->ev_last_valid = true;
  b->ev_valid = (e0 != nullptr && e0 == b->ev0);
  if (b->region_calls >= 0) b->region_calls++;
  b->last_stream = stream;
  b->nb_phase = (b->nb_phase + (uint32_t)(n_blocks % 3)) % 3u;
  b->als_phase = (b->als_phase + (uint32_t)n_blocks) & 1u;
  b->lo_parity ^= (uint32_t)(n_launch & 1);
  return 0;
}

int asdr_update_device(asdr_batch_t *b, const int16_t *dI, const int16_t *dQ, int16_t *dOut, int n_blocks, void *stream_) {
  return asdr_update_device_strided(b, dI, dQ, dOut, n_blocks, n_blocks, n_blocks, stream_);
}

// ---- capture sink: each channel's audio appended to one contiguous HBM row ---------------
int asdr_capture_open(asdr_batch_t *b, long capacity_blocks) {
  if (!b) return fail("null batch");
  if (b->device == ASDR_NO_DEVICE) return fail("control-plane-only batch (ASDR_NO_DEVICE): the capture sink lives in HBM");
  if (capacity_blocks <= 0 || capacity_blocks > 0x7fffffffL) return fail("bad capture capacity");
  HIPCHK(hipSetDevice(b->device));
  if (b->d_capture) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(b->d_capture)); b->d_capture = nullptr; }
  const size_t bytes = (size_t)b->n * (size_t)capacity_blocks * ASDR_N * sizeof(int16_t);
  if (hipMalloc(&b->d_capture, bytes) != hipSuccess) { b->d_capture = nullptr; return fail("capture sink: out of device memory"); }
  b->capture_cap = capacity_blocks; b->capture_pos = 0;
  return 0;
}

int asdr_capture_close(asdr_batch_t *b) {
  if (!b) return fail("null batch");
  if (b->d_capture) { HIPCHK(hipSetDevice(b->device)); HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(b->d_capture)); }
  b->d_capture = nullptr; b->capture_cap = 0; b->capture_pos = 0;
  return 0;
}

long asdr_capture_capacity(const asdr_batch_t *b) { return b ? b->capture_cap : 0; }
long asdr_capture_position(const asdr_batch_t *b) { return b ? b->capture_pos : 0; }
int16_t *asdr_capture_device_ptr(asdr_batch_t *b) { return b ? b->d_capture : nullptr; }
int asdr_capture_rewind(asdr_batch_t *b) { if (!b) return fail("null batch"); b->capture_pos = 0; return 0; }

int asdr_capture_update_device(asdr_batch_t *b, const int16_t *dI, const int16_t *dQ, int n_blocks, long in_stride_blocks, void *stream) {
  if (!b) return fail("null batch");
  if (!b->d_capture) return fail("capture sink is not open");
  if (!dI || !dQ) return 0;   // missing input: nothing is transmitted, nothing is appended (AudioSDR.cpp:48-56)
  if (n_blocks <= 0) return 0;
  if (b->capture_pos + n_blocks > b->capture_cap) return fail("capture sink overflow");
  int16_t *dst = b->d_capture + (size_t)b->capture_pos * ASDR_N;
  if (asdr_update_device_strided(b, dI, dQ, dst, n_blocks, in_stride_blocks, b->capture_cap, stream) != 0) return -1;
  b->capture_pos += n_blocks;
  return 0;
}

int asdr_capture_read(asdr_batch_t *b, int ch, long first_block, long n_blocks, int16_t *host_out) {
  if (!b) return fail("null batch");
  if (!b->d_capture) return fail("capture sink is not open");
  if (ch < 0 || ch >= b->n) return fail("bad channel");
  if (first_block < 0 || n_blocks < 0 || first_block + n_blocks > b->capture_pos) return fail("capture read beyond the write position");
  if (n_blocks == 0) return 0;
  if (!host_out) return fail("null output");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->last_stream));   // nullptr = the null stream
  HIPCHK(hipMemcpy(host_out, b->d_capture + ((size_t)ch * b->capture_cap + first_block) * ASDR_N,
                   (size_t)n_blocks * ASDR_N * sizeof(int16_t), hipMemcpyDeviceToHost));
  return 0;
}

int asdr_update(asdr_batch_t *b, const int16_t *I, const int16_t *Q, int16_t *out, int n_blocks) {
  if (!b) return fail("null batch");
  if (b->device == ASDR_NO_DEVICE) return fail("control-plane-only batch (ASDR_NO_DEVICE): the signal path needs a HIP device");
  if (!I || !Q) return 0;
  if (!out) return fail("null output");
  if (n_blocks <= 0) return 0;
  HIPCHK(hipSetDevice(b->device));
  const size_t count = (size_t)b->n * n_blocks * ASDR_N;
  if (count > b->io_cap) {
    HIPCHK(hipStreamSynchronize(b->stream));
    for (int i = 0; i < 3; i++) {
      if (b->d_io[i]) HIPCHK(hipFree(b->d_io[i]));
      b->d_io[i] = nullptr;
      HIPCHK(hipMalloc(&b->d_io[i], count * sizeof(int16_t)));
    }
    b->io_cap = count;
  }
  HIPCHK(hipMemcpyAsync(b->d_io[0], I, count * sizeof(int16_t), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_io[1], Q, count * sizeof(int16_t), hipMemcpyHostToDevice, b->stream));
  if (asdr_update_device(b, b->d_io[0], b->d_io[1], b->d_io[2], n_blocks, b->stream) != 0) return -1;
  HIPCHK(hipMemcpyAsync(out, b->d_io[2], count * sizeof(int16_t), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return check_stream_error(b);
}

int asdr_synchronize(asdr_batch_t *b) {
  if (!b) return fail("null batch");
  if (b->device == ASDR_NO_DEVICE) return 0;
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->last_stream));   // nullptr = the null stream
  HIPCHK(hipStreamSynchronize(b->stream));
  return check_stream_error(b);
}

float asdr_last_kernel_ms(asdr_batch_t *b) {
  if (!b || !b->ev_valid) return -1.0f;
  float ms = -1.0f;
  if (hipEventSynchronize(b->ev1) != hipSuccess) return -1.0f;
  if (hipEventElapsedTime(&ms, b->ev0, b->ev1) != hipSuccess) return -1.0f;
  return ms;
}

int asdr_set_launch_timing(asdr_batch_t *b, int on) {
  if (!b) return fail("null batch");
  b->time_calls = on != 0;
  if (!on) b->ev_valid = false;
  return 0;
}

int asdr_region_timing_begin(asdr_batch_t *b, void *stream_) {
  if (!b) return fail("null batch");
  if (b->device == ASDR_NO_DEVICE) return fail("control-plane-only batch has no device state");
  HIPCHK(hipSetDevice(b->device));
  b->region_stream = (hipStream_t)stream_;
  HIPCHK(hipEventRecord(b->rev0, b->region_stream));
  b->region_calls = 0;
  return 0;
}

int asdr_region_timing_end(asdr_batch_t *b, float *ms_total, long *n_calls) {
  if (!b) return fail("null batch");
  if (b->region_calls < 0) return fail("asdr_region_timing_end without asdr_region_timing_begin");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipEventRecord(b->rev1, b->region_stream));
  HIPCHK(hipEventSynchronize(b->rev1));
  float ms = 0.0f;
  HIPCHK(hipEventElapsedTime(&ms, b->rev0, b->rev1));
  if (ms_total) *ms_total = ms;
  if (n_calls) *n_calls = b->region_calls;
  b->region_calls = -1;
  return 0;
}

int asdr_kernel_timing_begin(asdr_batch_t *b, int max_launches) {
  if (!b) return fail("null batch");
  if (b->device == ASDR_NO_DEVICE) return fail("control-plane-only batch has no device state");
  HIPCHK(hipSetDevice(b->device));
  for (hipEvent_t e : b->tev) hipEventDestroy(e);
  b->tev.clear(); b->tev_used = 0;
  for (int i = 0; i < 2 * max_launches; i++) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); b->tev.push_back(e); }
  return 0;
}

int asdr_kernel_timing_end(asdr_batch_t *b, float *ms, int cap) {
  if (!b) return fail("null batch");
  const int n = (int)(b->tev_used / 2);
  for (int i = 0; i < n && i < cap; i++) {
    HIPCHK(hipEventSynchronize(b->tev[2 * i + 1]));
    HIPCHK(hipEventElapsedTime(&ms[i], b->tev[2 * i], b->tev[2 * i + 1]));
  }
  for (hipEvent_t e : b->tev) hipEventDestroy(e);
  b->tev.clear(); b->tev_used = 0;
  return n < cap ? n : cap;
}

// ---- general ------------------------------------------------------------------------------------------
void asdr_init(asdr_batch_t *b, int ch) { each(b, ch, [&](int i, Chan &) { chan_init(b, i); }); }
void asdr_setMute(asdr_batch_t *b, int ch, int muted) {  // .cpp:249-253
  each(b, ch, [&](int, Chan &c) { c.muted = muted != 0; c.current_out_gain = c.muted ? 0.0f : c.out_gain; });
}
int asdr_getMute(asdr_batch_t *b, int ch) { const Chan *c = get(b, ch); return c ? c->muted : 0; }
void asdr_setInputGain(asdr_batch_t *b, int ch, float g) {  // .cpp:232-238
  if (g > 10.0) g = 10.0f;
  if (g < 0.0) g = 0.0f;
  each(b, ch, [&](int, Chan &c) { c.in_gain = g; c.in_gain_i = c.in_gain * c.gain_balance; c.in_gain_q = c.in_gain; });
}
void asdr_setIQgainBalance(asdr_batch_t *b, int ch, float balance) {  // .cpp:240-244 (a local shadows _gainBalance)
  const float gb = sqrtf(balance);
  each(b, ch, [&](int, Chan &c) { c.in_gain_i = c.in_gain * gb; c.in_gain_q = c.in_gain / gb; });
}
void asdr_setOutputGain(asdr_batch_t *b, int ch, float g) { each(b, ch, [&](int, Chan &c) { c.output_gain = g; }); }
float asdr_setDemodMode(asdr_batch_t *b, int ch, int mode) {
  float r = 0.0f;
  bool first = true;
  each(b, ch, [&](int i, Chan &) { float v = set_mode(b, i, mode); if (first) { r = v; first = false; } });
  return r;
}
int16_t asdr_getDemodMode(asdr_batch_t *b, int ch) { const Chan *c = get(b, ch); return c ? (int16_t)c->mode : 0; }
float asdr_getTuningOffset(asdr_batch_t *b, int ch) { const Chan *c = get(b, ch); return c ? c->freq_shift : 0.0f; }
float asdr_getBPFlower(asdr_batch_t *b, int ch) {  // .cpp:259-265
  const Chan *c = get(b, ch);
  if (!c) return 0.0f;
  const uint16_t m = c->mode;
  if (m == ASDR_USBmode || m == ASDR_LSBmode) return (float)((double)kIFcenter - (double)kBWssb / 2.0);
  else if (m == ASDR_CW_USBmode || m == ASDR_CW_LSBmode) return (float)((double)kIFcenter - (double)kBWcw / 2.0);
  else if (m == ASDR_AMmode || m == ASDR_SAMmode) return (float)((double)kIFcenter - (double)kBWam / 2.0);
  else if (m == ASDR_WSPRmode) return (float)((double)kIFcenter - (double)kBWwspr / 2.0);
  return 0.0f;
}
float asdr_getBPFupper(asdr_batch_t *b, int ch) {  // .cpp:267-273 (WSPR: `+-` at :271)
  const Chan *c = get(b, ch);
  if (!c) return 0.0f;
  const uint16_t m = c->mode;
  if (m == ASDR_USBmode || m == ASDR_LSBmode) return (float)((double)kIFcenter + (double)kBWssb / 2.0);
  else if (m == ASDR_CW_USBmode || m == ASDR_CW_LSBmode) return (float)((double)kIFcenter + (double)kBWcw / 2.0);
  else if (m == ASDR_AMmode || m == ASDR_SAMmode) return (float)((double)kIFcenter + (double)kBWam / 2.0);
  else if (m == ASDR_WSPRmode) return (float)((double)kIFcenter + -((double)kBWwspr / 2.0));
  return 0.0f;
}

// ---- audio filter ----------------------------------------------------------------------------------------
void asdr_enableAudioFilter(asdr_batch_t *b, int ch) { each(b, ch, [&](int, Chan &c) { c.af_en = true; }); }
void asdr_disableAudioFilter(asdr_batch_t *b, int ch) { each(b, ch, [&](int, Chan &c) { c.af_en = false; }); }
void asdr_setAudioFilter(asdr_batch_t *b, int ch, int filter) {  // .cpp:298-311
  each(b, ch, [&](int i, Chan &c) {
    if (filter == ASDR_audioBypass) c.af_en = false;
    else if (filter >= ASDR_audioAM && filter <= ASDR_audio3300) {
      c.audio_table = ASDR_TBL_AUDIO_BASE + filter;
      mark_reset(b, i, ASDR_R_AF);
    }
    c.current_filter = (int16_t)filter;
  });
}
int asdr_getAudioFilter(asdr_batch_t *b, int ch) { const Chan *c = get(b, ch); return c ? c->current_filter : 0; }

// ---- ALS ---------------------------------------------------------------------------------------------------
void asdr_enableALSfilter(asdr_batch_t *b, int ch) {  // .cpp:384-391
  each(b, ch, [&](int i, Chan &c) { c.als_en = true; mark_reset(b, i, ASDR_R_ALS); });
}
void asdr_disableALSfilter(asdr_batch_t *b, int ch) { each(b, ch, [&](int, Chan &c) { c.als_en = false; }); }
void asdr_setALSfilterNotch(asdr_batch_t *b, int ch) { each(b, ch, [&](int, Chan &c) { c.als_notch = true; }); }
void asdr_setALSfilterPeak(asdr_batch_t *b, int ch) { each(b, ch, [&](int, Chan &c) { c.als_notch = false; }); }
void asdr_setALSfilterAdaptive(asdr_batch_t *b, int ch) { each(b, ch, [&](int, Chan &c) { c.als_adaptive = true; }); }
void asdr_setALSfilterStatic(asdr_batch_t *b, int ch) { each(b, ch, [&](int, Chan &c) { c.als_adaptive = false; }); }
void asdr_setALSfilterParams(asdr_batch_t *b, int ch, unsigned int m, float lambda, float delay) {  // .cpp:393-398
  each(b, ch, [&](int, Chan &c) {
    c.als_m = (int16_t)m;
    if (c.als_m >= ASDR_N) c.als_m = ASDR_N;
    c.als_lambda = lambda;
    c.als_delay = (int16_t)delay;
  });
}
int asdr_ALSfilterIsEnabled(asdr_batch_t *b, int ch) { const Chan *c = get(b, ch); return c ? c->als_en : 0; }
int asdr_ALSfilterIsNotch(asdr_batch_t *b, int ch) { const Chan *c = get(b, ch); return c ? c->als_notch : 0; }
int asdr_ALSfilterIsPeak(asdr_batch_t *b, int ch) { const Chan *c = get(b, ch); return c ? !c->als_notch : 0; }
int asdr_ALSfilterIsAdaptive(asdr_batch_t *b, int ch) { const Chan *c = get(b, ch); return c ? c->als_adaptive : 0; }

// ---- AGC ---------------------------------------------------------------------------------------------------
void asdr_enableAGC(asdr_batch_t *b, int ch) { each(b, ch, [&](int, Chan &c) { c.agc_en = true; }); }
void asdr_disableAGC(asdr_batch_t *b, int ch) { each(b, ch, [&](int, Chan &c) { c.agc_en = false; }); }
int asdr_AGCisEnabled(asdr_batch_t *b, int ch) { const Chan *c = get(b, ch); return c ? c->agc_en : 0; }
void asdr_setAGCthreshold(asdr_batch_t *b, int ch, float v) { each(b, ch, [&](int, Chan &c) { c.agc_threshold = v; rebuild_agc(b, c); }); }
void asdr_setAGCslope(asdr_batch_t *b, int ch, float v) { each(b, ch, [&](int, Chan &c) { c.agc_slope = v; rebuild_agc(b, c); }); }
void asdr_setAGCkneeWidth(asdr_batch_t *b, int ch, float v) { each(b, ch, [&](int, Chan &c) { c.agc_knee = v; rebuild_agc(b, c); }); }
void asdr_setAGCattackTime(asdr_batch_t *b, int ch, float ms) {  // .cpp:551-555
  const float al = time_constant(ms), be = (float)(1.0 - (double)al);
  each(b, ch, [&](int, Chan &c) { c.agc_attack_ms = ms; c.agc_alpha_att = al; c.agc_beta_att = be; });
}
void asdr_setAGCreleaseTime(asdr_batch_t *b, int ch, float ms) {  // .cpp:557-561
  const float al = time_constant(ms), be = (float)(1.0 - (double)al);
  each(b, ch, [&](int, Chan &c) { c.agc_release_ms = ms; c.agc_alpha_rel = al; c.agc_beta_rel = be; });
}
void asdr_setAGChangTime(asdr_batch_t *b, int ch, float ms) {  // .cpp:563-566: float product, then / 1000.0
  const uint32_t cnt = (uint32_t)((double)(ms * kFs) / 1000.0);
  each(b, ch, [&](int, Chan &c) { c.agc_slot129 = ms; c.agc_hang_count = cnt; });
}
void asdr_setAGCstaticGain(asdr_batch_t *b, int ch, float g) { each(b, ch, [&](int, Chan &c) { c.agc_static_gain = g; }); }
void asdr_setAGCmode(asdr_batch_t *b, int ch, int mode) {  // .cpp:524-544
  mode = (int16_t)mode;
  if (mode == ASDR_AGCoff) asdr_disableAGC(b, ch);
  else if (mode == ASDR_AGCfast) { asdr_setAGCattackTime(b, ch, 2.0f); asdr_setAGCreleaseTime(b, ch, 100.0f); asdr_setAGChangTime(b, ch, 100.0f); asdr_enableAGC(b, ch); }
  else if (mode == ASDR_AGCmedium) { asdr_setAGCattackTime(b, ch, 5.0f); asdr_setAGCreleaseTime(b, ch, 250.0f); asdr_setAGChangTime(b, ch, 500.0f); asdr_enableAGC(b, ch); }
  else if (mode == ASDR_AGCslow) { asdr_setAGCattackTime(b, ch, 10.0f); asdr_setAGCreleaseTime(b, ch, 500.0f); asdr_setAGChangTime(b, ch, 2000.0f); asdr_enableAGC(b, ch); }
}
#define GETF(name, field) float asdr_##name(asdr_batch_t *b, int ch) { const Chan *c = get(b, ch); return c ? c->field : 0.0f; }
GETF(getAGCthreshold, agc_threshold)
GETF(getAGCslope, agc_slope)
GETF(getAGCkneeWidth, agc_knee)
GETF(getAGCattack, agc_attack_ms)
GETF(getAGCrelease, agc_release_ms)
GETF(getAAGalphaAttack, agc_alpha_att)
GETF(getAGCbetaAttack, agc_beta_att)
GETF(getAGCalphaRelease, agc_alpha_rel)
GETF(getAGCbetaRelease, agc_beta_rel)
GETF(getAGCstaticGain, agc_static_gain)
float asdr_getAGClookup(asdr_batch_t *b, int ch, int i) {
  const Chan *c = get(b, ch);
  if (!c || i < 0 || i > 129) return 0.0f;
  return (i == 129) ? c->agc_slot129 : b->agc_pool[c->agc_table].v[i];
}

// ---- noise blanker ------------------------------------------------------------------------------------------
void asdr_enableNoiseBlanker(asdr_batch_t *b, int ch) {
  each(b, ch, [&](int i, Chan &c) { c.nb_en = true; mark_reset(b, i, ASDR_R_NB); });
}
void asdr_disableNoiseBlanker(asdr_batch_t *b, int ch) { each(b, ch, [&](int, Chan &c) { c.nb_en = false; }); }
void asdr_setNoiseBlankerThreshold(asdr_batch_t *b, int ch, float r) {
  each(b, ch, [&](int i, Chan &c) { c.nb_threshold = r; mark_reset(b, i, ASDR_R_NB); });
}
void asdr_setNoiseBlankerThresholdDb(asdr_batch_t *b, int ch, float db) {  // .cpp:671-674
  const float r = powf(10.0f, (float)((double)db / 20.0));
  each(b, ch, [&](int i, Chan &c) { c.nb_threshold = r; mark_reset(b, i, ASDR_R_NB); });
}
int asdr_NoiseBlankerisEnabled(asdr_batch_t *b, int ch) { const Chan *c = get(b, ch); return c ? c->nb_en : 0; }

// ---- getters that read hot-path state ------------------------------------------------------------------------
int asdr_AGCisActive(asdr_batch_t *b, int ch) { ChanSmall s; return read_small(b, ch, s) == 0 ? ((s.status & ASDR_S_AGC_ACTIVE) != 0) : 0; }
int asdr_NoiseBlankerDetection(asdr_batch_t *b, int ch) { ChanSmall s; return read_small(b, ch, s) == 0 ? ((s.status & ASDR_S_NB_DETECTED) != 0) : 0; }
int asdr_getSAMphaseLockStatus(asdr_batch_t *b, int ch) { ChanSmall s; return read_small(b, ch, s) == 0 ? ((s.status & ASDR_S_PLL_LOCKED) != 0) : 0; }
float asdr_getSAMfrequency(asdr_batch_t *b, int ch) { ChanSmall s; return read_small(b, ch, s) == 0 ? s.pll_freq : 0.0f; }
float asdr_getAMcarrierLevel(asdr_batch_t *b, int ch) { ChanSmall s; return read_small(b, ch, s) == 0 ? s.am_carrier : 0.0f; }

int asdr_read_status(asdr_batch_t *b, int32_t *agc_active, int32_t *nb_detected, int32_t *sam_locked, float *sam_frequency,
                     float *am_carrier) {
  if (!b) return fail("null batch");
  ChanSmall probe;
  if (read_small(b, 0, probe) != 0) return -1;  // synchronises + applies pending resets
  std::vector<ChanSmall> all(b->n);
  HIPCHK(hipMemcpy(all.data(), b->d_small, (size_t)b->n * sizeof(ChanSmall), hipMemcpyDeviceToHost));
  for (int i = 0; i < b->n; i++) {
    if (agc_active) agc_active[i] = (all[i].status & ASDR_S_AGC_ACTIVE) != 0;
    if (nb_detected) nb_detected[i] = (all[i].status & ASDR_S_NB_DETECTED) != 0;
    if (sam_locked) sam_locked[i] = (all[i].status & ASDR_S_PLL_LOCKED) != 0;
    if (sam_frequency) sam_frequency[i] = all[i].pll_freq;
    if (am_carrier) am_carrier[i] = all[i].am_carrier;
  }
  return 0;
}

unsigned int asdr_get_chain_constants(asdr_batch_t *b, int ch, float out[12]) {
  if (!b) return 0;
  const ChainConsts &k = b->k;
  if (out) {
    const float v[12] = {k.pll_b0, k.pll_b1, k.pll_a1, k.pll_alpha_freq, k.pll_beta_freq, k.pll_f_conv, k.pll_lock_lo, k.pll_lock_hi,
                         k.two_pi_f, k.half_pi_f, k.phase_inc_unit, k.nb_beta};
    memcpy(out, v, sizeof v);
  }
  const Chan *c = get(b, ch);
  return c ? c->agc_hang_count : 0u;
}

long asdr_stream_pipeline_launches(asdr_batch_t *b) { return b ? b->stat_stream_launches : -1; }
long asdr_stream_pipeline_recoveries(asdr_batch_t *b) {
  if (!b) return -1;
  if (b->device != ASDR_NO_DEVICE && b->stream_launched) { if (hipSetDevice(b->device) != hipSuccess || check_stream_error(b) != 0) return -1; }
  return b->stat_stream_recoveries;
}
int asdr_stream_pipeline_max_groups(asdr_batch_t *b) { return b ? b->stream_max_waves : -1; }
int asdr_set_stream_pipeline(asdr_batch_t *b, int on) { if (!b) return fail("null batch"); b->stream_pipeline = on != 0; return 0; }
int asdr_set_als_launch_form(asdr_batch_t *b, int split_min_channels) {
  if (!b) return fail("null batch");
  b->als_split_min = split_min_channels > 0 ? split_min_channels : 0x7fffffff;
  b->sched_dirty = true;
  return 0;
}
int asdr_set_sam_launch_form(asdr_batch_t *b, int fused, int split_min_channels) {
  if (!b) return fail("null batch");
  b->sam_fused = fused != 0;
  b->sam_split_min = split_min_channels > 0 ? split_min_channels : ASDR_SAM_SPLIT_MIN_CHANNELS;
  b->sched_dirty = true;
  return 0;
}
int asdr_debug_set_stream_max_groups(asdr_batch_t *b, int groups) {   // experiments: more than one pipeline workgroup per compute unit
  if (!b) return fail("null batch");
  if (groups < 0 || groups > 1024) return fail("bad group count");
  b->stream_max_waves = groups;
  return 0;
}
int asdr_debug_set_stream_spin_limit(asdr_batch_t *b, unsigned int polls) {
  if (!b) return fail("null batch");
  b->stream_spin_limit = polls ? polls : ASDR_STREAM_SPIN_LIMIT;
  return 0;
}

int asdr_set_exact_unknown_mode(asdr_batch_t *b, int on) {
  if (!b) return fail("null batch");
  const bool want = on != 0;
  if (want && !b->exact_unknown_mode && b->device != ASDR_NO_DEVICE) {   // rows not kept meanwhile: silence until a block stores them again
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->last_stream));
    HIPCHK(hipMemset(b->d_audio_prev, 0, ((size_t)b->n + 1) * 128 * sizeof(float)));
  }
  b->exact_unknown_mode = want;
  return 0;
}
int asdr_get_exact_unknown_mode(asdr_batch_t *b) { return b ? (b->exact_unknown_mode ? 1 : 0) : -1; }

int asdr_schedule_layout(asdr_batch_t *b, int out[8]) {
  if (!b || !out) return fail("null argument");
  for (int k = 0; k < ASDR_KERNEL_KINDS; k++) out[k] = b->kind_slots[k];
  out[5] = b->left_slots; out[6] = b->left_slots ? b->left_kind : -1; out[7] = (b->sam_split ? 1 : 0) | (b->als_split ? 2 : 0);
  return 0;
}

int asdr_control_plane_flush(asdr_batch_t *b, long long stats[4]) {
  if (!b) return fail("null batch");
  if (b->device != ASDR_NO_DEVICE) return fail("asdr_control_plane_flush is for control-plane-only batches: a device batch flushes in update()");
  std::vector<int32_t> changed;
  bool bulk = false, rebuilt = false;
  flush_host(b, changed, bulk, rebuilt);
  b->agc_pool_dirty = false;
  if (stats) {
    stats[0] = b->stat_rows_refilled; stats[1] = rebuilt ? 1 : 0;
    const int lw = b->left_slots / 8;   // the remainders' waves count with the kind whose general kernel runs them
    stats[2] = (long long)(b->kind_slots[ASDR_KERNEL_PLAIN] / 8 + (b->left_kind == ASDR_KERNEL_PLAIN ? lw : 0)) | ((long long)(b->kind_slots[ASDR_KERNEL_SAM] / 8) << 21) |
               ((long long)((b->kind_slots[ASDR_KERNEL_ALS] + b->kind_slots[ASDR_KERNEL_ALS_SMALL] + b->kind_slots[ASDR_KERNEL_SAM_ALS]) / 8 + (b->left_kind == ASDR_KERNEL_ALS ? lw : 0)) << 42);
    long long live = 0;
    for (uint32_t r : b->agc_refs) live += (r != 0);
    stats[3] = live;
  }
  return 0;
}

int asdr_enable_taps(asdr_batch_t *b, int on) {
  if (!b) return fail("null batch");
  if (b->device == ASDR_NO_DEVICE) return fail("control-plane-only batch has no device state");
  HIPCHK(hipSetDevice(b->device));
  if (on && !b->d_taps) {
    const size_t bytes = (size_t)ASDR_N_TAPS * b->n * ASDR_N * sizeof(float);
    HIPCHK(hipMalloc(&b->d_taps, bytes));
    HIPCHK(hipMemset(b->d_taps, 0, bytes));
  }
  b->taps_on = on != 0;
  return 0;
}

int asdr_read_taps(asdr_batch_t *b, float *dst) {
  if (!b || !b->d_taps || !dst) return fail("taps not enabled");
  if (asdr_synchronize(b) != 0) return -1;
  HIPCHK(hipMemcpy(dst, b->d_taps, (size_t)ASDR_N_TAPS * b->n * ASDR_N * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
