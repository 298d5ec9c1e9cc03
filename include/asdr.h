/* asdr.h -- C ABI of libasdr_hip.so: batched AudioSDR::update() on AMD Instinct MI355X (gfx950).
 *
 * One `asdr_batch_t` holds N independent channels; one channel == one instance of the reference's
 * `class AudioSDR : public AudioStream` (/root/reference/SRC/AudioSDRlib/AudioSDR.h:75-156).
 * Every entry point below names the reference interface it replaces.  Plain pointers and sizes
 * only; no C++ or torch types.  There is NO CPU path: the library needs a HIP device.
 *
 * Conventions
 *  - `ch` selects a channel 0..N-1; `ch == ASDR_ALL` (-1) applies a setter to every channel.
 *    Getters need a real channel index.
 *  - Setters take effect at the next asdr_update*() call (the reference's setters run in loop()
 *    context between audio interrupts; SURVEY.md 5 "race detection").
 *  - Enumerators keep the reference's integer values (AudioSDR.h:44-71).
 *  - Status getters that read hot-path state (AGCisActive, NoiseBlankerDetection, SAM lock/frequency,
 *    AM carrier level) synchronise with the batch's stream and read device memory.
 *  - Return codes: 0 = ok, <0 = error (asdr_last_error() gives text).  Setters mirror the
 *    reference's void/float returns and its (non-)clamping.
 *  - Not thread-safe per batch; use one batch per GPU / per host thread.
 */
#ifndef ASDR_H_
#define ASDR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASDR_BLOCK_SAMPLES 128 /* AUDIO_BLOCK_SAMPLES; AudioSDR.h:73 n_block */
#define ASDR_ALL (-1)
#define ASDR_NO_DEVICE (-1) /* asdr_create(): control plane only (setters/getters); every update fails */

/* demodulation modes, AudioSDR.h:44-50 */
enum { ASDR_LSBmode = 0, ASDR_USBmode = 1, ASDR_CW_LSBmode = 2, ASDR_CW_USBmode = 3, ASDR_AMmode = 4,
       ASDR_SAMmode = 5, ASDR_WSPRmode = 6 };
/* audio filters, AudioSDR.h:56-66 */
enum { ASDR_audioAM = 0, ASDR_audioCW = 1, ASDR_audioWSPR = 2, ASDR_audio2100 = 3, ASDR_audio2300 = 4,
       ASDR_audio2500 = 5, ASDR_audio2700 = 6, ASDR_audio2900 = 7, ASDR_audio3100 = 8, ASDR_audio3300 = 9,
       ASDR_audioBypass = 10 };
/* AGC presets, AudioSDR.h:68-71 */
enum { ASDR_AGCoff = 0, ASDR_AGCfast = 1, ASDR_AGCmedium = 2, ASDR_AGCslow = 3 };

typedef struct asdr_batch asdr_batch_t;

/* ---- lifetime: replaces N x `AudioSDR sdr;` (constructor AudioSDR.h:77-79 -> init() AudioSDR.cpp:174-185).
 * `device` is a HIP device ordinal, or ASDR_NO_DEVICE for a control-plane-only batch whose setters and
 * pure-parameter getters work but whose update()/status getters fail (there is no CPU signal path).
 * Returns NULL on failure. */
asdr_batch_t *asdr_create(int n_channels, int device);
void asdr_destroy(asdr_batch_t *b);
const char *asdr_last_error(void);
int asdr_n_channels(const asdr_batch_t *b);

/* ---- sharded batches: replaces N x `AudioSDR sdr;` spread over the GPUs of one node (instances are independent,
 * AudioSDR.h:75-81, AudioSDR.cpp:41-44).  Shard g of n_shards owns channels [g*C/n_shards, (g+1)*C/n_shards) -- state, parameters,
 * schedule, streams and staging all live on devices[g] (ordinals may repeat: several shards on one GPU; ASDR_NO_DEVICE for every
 * shard = a control-plane-only sharded batch).  The handle is an asdr_batch_t: EVERY entry point of this header takes it, with
 * GLOBAL channel indices -- setters and getters are routed to the owner (ASDR_ALL fans out), status / tap / capture reads come back
 * in global channel order -- so code written for one batch runs unchanged on eight GPUs.  There is no collective and nothing crosses
 * between devices.
 *   asdr_update              host rows of all channels: scattered to / gathered from the shards' overlapped host paths, one host
 *                            thread per shard.
 *   asdr_update_device*,     device pointers belong to one device: accepted while all shards live on the SAME device (each shard
 *   asdr_capture_update_device  takes its rows of the caller's arrays); for shards on several devices drive each shard with its own
 *                            device-local pointers and stream through asdr_shard(b, g) -- the calls are asynchronous, so one host
 *                            thread can keep all GPUs busy.
 *   asdr_region_timing_*, asdr_kernel_timing_*, asdr_capture_device_ptr   per device: through the shard handles.
 * asdr_shard_first_channel(b, g) = first global channel of shard g (g == n_shards: the total).  A plain batch answers as one shard. */
asdr_batch_t *asdr_create_sharded(int n_channels, int n_shards, const int *devices);
int asdr_n_shards(const asdr_batch_t *b);
asdr_batch_t *asdr_shard(asdr_batch_t *b, int shard);           /* owned by the sharded batch: never asdr_destroy() it */
int asdr_shard_first_channel(const asdr_batch_t *b, int shard);
int asdr_shard_device(const asdr_batch_t *b, int shard);

/* ---- the hot path: replaces AudioSDR::update() (AudioSDR.cpp:39-168), called once per 128-sample
 * block per instance by the Teensy audio interrupt with two int16 blocks in (receiveWritable(0/1),
 * :46-47) and one mono int16 block out (transmit, :164-165).
 * Layout of I, Q and out: [channel][block][128] int16, contiguous (row = 256 B).  n_blocks >= 1
 * consecutive blocks per channel are processed in order in one call.
 *  asdr_update        : host pointers; copies in, runs, copies out, synchronises.
 *  asdr_update_device : device pointers (already resident in HBM); asynchronous on `stream`
 *                       (a hipStream_t; NULL = the null stream).  Buffers must stay valid until
 *                       the stream reaches this point.  Calls may come on different streams: a call on
 *                       another stream than the previous one first waits (event) for that call's kernels,
 *                       so the previous stream must still exist when the next call is made.
 * A NULL I or Q mirrors the reference's missing-input guard (AudioSDR.cpp:48-56): nothing is
 * processed, no state advances, out is untouched, return 0. */
int asdr_update(asdr_batch_t *b, const int16_t *I, const int16_t *Q, int16_t *out, int n_blocks);
/* asdr_update() is the boundary the reference's data path has (host-resident audio blocks in, one out: AudioSDR.cpp:46-47, 158-167):
 * 768 bytes cross PCIe per channel-block and that, not the kernels, bounds the call.  It is overlapped: the batch is cut into
 * channel-range chunks and H2D(k + 1) || kernels(k) || D2H(k - 1) run on three streams (chunking is by channels, so the order of
 * blocks within a channel is untouched and results are bit-identical to asdr_update_device).  Caller buffers that are PINNED are DMA
 * targets as they are; pageable buffers go through a pinned staging area of the batch, copied by a few worker threads
 * (ASDR_HOST_COPY_THREADS, default 4).  To pin:
 *   asdr_host_alloc / asdr_host_free          page-locked host memory (hipHostMalloc, portable across devices)
 *   asdr_host_register / asdr_host_unregister pin memory the caller already owns, in place (hipHostRegister); unregister before freeing it
 *   asdr_set_host_chunks(b, k)                k > 0 forces k chunks (1 = no overlap: the round-3 behaviour, for comparison), 0 = from the call's size
 *   asdr_host_path_info(b, out)               out[0] = chunks of the last asdr_update, out[1] = 1 if it used the caller's buffers directly
 * Pageable buffers that RECUR can be pinned for the caller (round 6, OPT-IN): with asdr_host_autopin(1) the second time asdr_update is
 * handed the same (address, length) of ordinary memory it registers the range in place, and every later call DMA-copies straight from /
 * into it -- an application that reuses its buffers, as the reference's audio library does, pays the staging copies once (C2: 1.29 ->
 * 1.06 ms per call).  At most 12 ranges / 2 GiB are kept registered (least recently used first out -- never a range that a call, of any thread, is copying
 * through at that moment: a call holds its ranges for its length, and a new range that finds nothing evictable stays on the staged path);
 * all of them are released when the process' last batch is destroyed.  THE CONTRACT that makes it opt-in: a buffer the cache holds must not be freed (unmapped) behind the
 * library's back -- a DMA through a registration whose range was unmapped and mapped again aborts the process (measured), and nothing
 * tells a library that its caller unmapped a range.  Call asdr_host_autopin_clear() before freeing buffers you have passed in.
 * AND A WARNING beyond the contract (round 6, measured): a long-lived process that had registered and correctly released many SMALL ordinary allocations
 * this way (heap blocks that share pages with their neighbours) and then went on allocating, freeing and copying other memory aborted inside the
 * runtime in a later, unrelated copy -- about every second run of this repository's GPU suite, until the tests that switch this on were given
 * processes of their own.  Use it for large, long-lived, page-exclusive buffers (the case it was built for: a receiver bank's I/Q rows) or pin them
 * yourself (asdr_host_alloc / asdr_host_register), which has no such history.
 *   asdr_host_autopin(on)                     1 / 0 switches the behaviour on / off for the process (default off; environment ASDR_HOST_AUTOPIN=1: on),
 *                                             -1 only asks; returns the previous setting
 *   asdr_host_autopin_clear()                 unregister everything the cache holds
 *   asdr_host_autopin_info(out)               out[0] = ranges registered now, out[1] = registrations so far, out[2] = released, out[3] = refused by the driver */
void *asdr_host_alloc(size_t bytes);
void asdr_host_free(void *p);
int asdr_host_register(void *p, size_t bytes);
int asdr_host_unregister(void *p);
int asdr_set_host_chunks(asdr_batch_t *b, int chunks);
int asdr_host_path_info(asdr_batch_t *b, int out[2]);
int asdr_host_autopin(int on);
void asdr_host_autopin_clear(void);
int asdr_host_autopin_info(long out[4]);
/* Test hooks (host logic only; they work on a control-plane-only batch after asdr_control_plane_flush): the chunk plan of the overlapped
 * host path for `chunks` chunks -- bound[0..chunks] channel boundaries, need_in[p] = the last input chunk kernel part p waits for,
 * last_part[j] = the part after which output chunk j is complete --, the schedule slots (first, count pairs, up to 16; returns their
 * number) that part `part` of `parts` launches, and the channel of every schedule slot (returns the slot count; n_channels = padding). */
int asdr_debug_host_plan(asdr_batch_t *b, int chunks, int *bound, int *need_in, int *last_part);
int asdr_debug_part_slots(asdr_batch_t *b, int part, int parts, int *out);
int asdr_debug_schedule(asdr_batch_t *b, int *channels, int cap);
int asdr_update_device(asdr_batch_t *b, const int16_t *dI, const int16_t *dQ, int16_t *dOut, int n_blocks,
                       void *stream);
/* Same, with explicit row strides in blocks: I and Q rows are in_stride_blocks*128 samples apart, out rows
 * out_stride_blocks*128 (both >= n_blocks).  Lets a caller stream out of a longer input buffer and into a longer
 * per-channel audio row without repacking.  All three pointers must be 16-byte aligned. */
int asdr_update_device_strided(asdr_batch_t *b, const int16_t *dI, const int16_t *dQ, int16_t *dOut, int n_blocks,
                               long in_stride_blocks, long out_stride_blocks, void *stream);
/* `stream` = ASDR_STREAM_BATCH: the call runs on streams of the batch's own and is ordered ONLY against the batch's other calls
 * (and against host-side synchronisation: asdr_synchronize, the status / capture readers).  Ordering against the caller's streams
 * is the caller's, through
 *   asdr_order_after(b, s)    the batch's later ASDR_STREAM_BATCH calls run after everything enqueued on s so far (inputs produced there)
 *   asdr_order_before(b, s)   everything enqueued on s from now on runs after the batch's calls so far (results consumed there)
 * What that buys: kernels of one stream run strictly one after the other, and each ends with a tail in which its last waves drain
 * and the GPU runs empty.  A batch whose schedule is one settings group of consecutive channels (every channel configured alike:
 * BASELINE configs 2 and 5) runs its two halves on two such streams ("lanes") that never wait for each other: half A of block k + 1
 * starts when half A of block k is done, while half B of block k still drains -- 0.121 -> 0.111 ms per step of 65,536 channels
 * (tools/split_probe.py).  On a caller's stream the same happens INSIDE a multi-block call of a large batch (issued as one launch per
 * block and joined at its end); from call to call it cannot, because the caller may have enqueued anything in between.  Results are
 * bit-identical either way (the halves touch disjoint channels; each lane has its own local-oscillator cache entries).
 * asdr_lane_calls() = calls that ran on the lanes so far; asdr_set_lanes(b, on, min_waves) switches them off / moves the smallest
 * batch (in waves of 8 channels, default 1024) that uses them; environment ASDR_NO_LANES=1 at asdr_create time = default off.
 * Since round 5 the pool's streams are created at the HIGHEST stream priority: the runtime keeps hardware queues per priority level, so the
 * pool does not share queues with streams the application creates at the default priority (measured: what follows, gone; environment
 * ASDR_POOL_PRIORITY=normal restores the default level, =low picks the lowest; asdr_set_pool_priority(0 normal / 1 highest / 2 lowest) does the
 * same from code, BEFORE the process' first asdr_create).  The price: every kernel the library launches on its own streams (lanes, role streams,
 * the host path's copies) is scheduled in front of the application's default-priority work on that device -- an application with
 * latency-critical kernels of its own beside the receiver bank may prefer the default level and shared hardware queues.
 * WHAT THE LANES REST ON, and what breaks them.  The batch's own streams come from ONE pool of three non-blocking streams per device and
 * process, shared by every batch of the process on that device (asdr_synchronize / the host path of one batch therefore also waits for
 * other batches' work on those streams; device ordinals 0..15).  HIP maps the streams a process uses onto a few hardware queues (four by
 * default: the null stream + the pool's three); two streams that share a queue run strictly one after the other, and every further
 * stream in the process -- the application's, a framework's -- can re-map the pool's streams onto a shared queue (measured: one
 * back-to-back kernel stream that shares its queue, 0.122 -> 0.153 ms per 65,536-channel step).  The first call that would use the lanes
 * therefore PROBES the pool (a 30-us spin kernel alone on the pool's first stream, then one on each of its first two streams behind a common
 * marker: the pair ends within 1.5 x the lone spin, or one after the other; three rounds, majority), and a batch whose pool streams do not
 * overlap stays on the ordinary path unless asdr_set_lanes(b, on > 0) asked for the lanes.  "Concurrent" is kept for the device and process;
 * "serialised" holds for the batch that asked -- the next batch's first lane-sized call probes again, and only the third such verdict is kept.
 * asdr_lanes_overlap_probe() = the last verdict: 1 concurrent / 0 serialised / -1 not probed yet (or ASDR_NO_LANES_PROBE=1, or the probe
 * failed: defaults kept), asdr_lanes_enabled() = what the batch's next lane-sized call will do.  (Not at asdr_create: a process that only
 * ever calls on its own streams never creates the pool's lane streams -- two application streams driving two shards of one GPU lost 25 %
 * to a probe at create time, round 5.)  The probe sees the stream population at that moment; an application that creates many streams
 * afterwards should re-check with its own timing, or keep its calls on its own stream (strict order, no lanes: bit-identical results,
 * ~10 % slower for large one-group batches). */
#define ASDR_STREAM_BATCH ((void *)(intptr_t)-1)
int asdr_lanes_overlap_probe(asdr_batch_t *b);
int asdr_set_pool_priority(int level);
int asdr_lanes_enabled(asdr_batch_t *b);
int asdr_order_after(asdr_batch_t *b, void *stream);
int asdr_order_before(asdr_batch_t *b, void *stream);
long asdr_lane_calls(asdr_batch_t *b);
/* SAM role streams: a multi-block call (n_blocks >= 2) of a batch whose schedule is ONE sub-range of SAM channels run as three launches
 * per block (pre | PLL | post: 512 or more SAM channels configured alike) puts the three roles on three streams chained by events, so
 * that pre(k + 1) and PLL(k + 1) run beside post(k): the PLL's 128-step dependent chain per block (about 32 us for any bank size) then
 * bounds the call alone -- 61 -> about 35 us per block for 512 receivers.  Results are bit-identical (the roles of a block touch disjoint
 * state; tiles and lock words alternate between two sets).  asdr_sam_role_calls() = calls that ran that way; environment
 * ASDR_NO_SAM_ROLE_STREAMS=1 at asdr_create time switches it off.  (AudioSDR.cpp:688-749 is the chain.) */
long asdr_sam_role_calls(asdr_batch_t *b);
/* ... of which (round 5) ran the three roles in CHUNKS of 8 blocks per launch (their block loops kept, 32 tile sets): uniform SAM banks
 * without the ALS filter of up to 4,096 channels, calls of 16 blocks or more -- one event pair per chunk and role instead of per block.
 * Environment ASDR_NO_SAM_CHUNKS=1 keeps the per-block form. */
long asdr_sam_chunk_calls(asdr_batch_t *b);
/* ALS role streams (round 5): a SMALL bank whose whole schedule is one settings group of channels with a short ALS filter (taps <= 64,
 * delay + taps <= 65; a known mode; below the one-launch-per-block size) runs a multi-block call as launches of 8 blocks per role on
 * event-chained streams -- by default THREE stages on three streams: front half (scale, blanker, IF band-pass) | back half (mixer .. AGC) | the
 * filter + output; ASDR_ALS_ROLE_STAGES=2: the chain up to the AGC | the filter + output on two --, a role's chunk beside the next role's
 * previous chunk.  The rows cross in HBM: a stage of 32 post-AGC rows per channel (16 KB per channel: 64 MB at 4,096 channels, allocated at
 * the first such call; a chain role may run up to two chunks ahead of the filter role) and, in the three-stage form, 32 tile sets between the
 * chain's halves (32 KB per channel: 128 MB at 4,096 channels, shared with the chunked SAM role streams).  Bit-identical to the block loop; not
 * taken for in-place calls or with stage taps.  asdr_als_role_calls() = calls that ran that way; environment ASDR_NO_ALS_ROLE_STREAMS=1 turns
 * it off. */
long asdr_als_role_calls(asdr_batch_t *b);
int asdr_set_lanes(asdr_batch_t *b, int on, int min_waves);
int asdr_synchronize(asdr_batch_t *b);

/* ---- capture sink (the continuous receive loop of EXTRAS/BareBonesWSPR/BareBonesWSPR.ino:
 * 87-135 with the audio kept instead of played): one contiguous mono int16 row per channel in HBM,
 * [channel][capacity_blocks*128] at 44.1 kHz, appended to by every asdr_capture_update_device() call, so that a
 * 2-minute WSPR slot (41,344 blocks) ends up as one row per receiver that a decoder can read in place
 * (asdr_capture_device_ptr) or copy out (asdr_capture_read).  The 12 kHz resampling WSPR decoders expect is not
 * part of the reference and is not done here. */
int asdr_capture_open(asdr_batch_t *b, long capacity_blocks);       /* (re)allocates, position = 0 */
int asdr_capture_close(asdr_batch_t *b);
long asdr_capture_capacity(const asdr_batch_t *b);                  /* blocks per channel row */
long asdr_capture_position(const asdr_batch_t *b);                  /* blocks appended so far */
int asdr_capture_rewind(asdr_batch_t *b);                           /* position = 0 (channel state is kept) */
int16_t *asdr_capture_device_ptr(asdr_batch_t *b);                  /* [n_channels][capacity_blocks][128] */
int asdr_capture_update_device(asdr_batch_t *b, const int16_t *dI, const int16_t *dQ, int n_blocks,
                               long in_stride_blocks, void *stream); /* update() x n_blocks, audio appended */
int asdr_capture_read(asdr_batch_t *b, int ch, long first_block, long n_blocks, int16_t *host_out);

/* ---- general (AudioSDR.h:88-97; AudioSDR.cpp:174-273) */
void asdr_init(asdr_batch_t *b, int ch);                              /* init()            .cpp:174 */
void asdr_setMute(asdr_batch_t *b, int ch, int muted);                /* setMute           .cpp:249 */
int asdr_getMute(asdr_batch_t *b, int ch);                            /* getMute           .cpp:255 */
void asdr_setInputGain(asdr_batch_t *b, int ch, float gain);          /* setInputGain      .cpp:232 */
void asdr_setIQgainBalance(asdr_batch_t *b, int ch, float balance);   /* setIQgainBalance  .cpp:240 */
void asdr_setOutputGain(asdr_batch_t *b, int ch, float gain);         /* setOutputGain     .cpp:245 */
float asdr_setDemodMode(asdr_batch_t *b, int ch, int mode);           /* setDemodMode      .cpp:187; returns the
                                                                         tuning offset (of channel 0 for ASDR_ALL) */
int16_t asdr_getDemodMode(asdr_batch_t *b, int ch);                   /* getDemodMode      .cpp:228 */
float asdr_getTuningOffset(asdr_batch_t *b, int ch);                  /* getTuningOffset   .cpp:224 */
float asdr_getBPFlower(asdr_batch_t *b, int ch);                      /* getBPFlower       .cpp:259 */
float asdr_getBPFupper(asdr_batch_t *b, int ch);                      /* getBPFupper       .cpp:267 */

/* ---- IIR audio output filter (AudioSDR.h:100-104; AudioSDR.cpp:289-311) */
void asdr_enableAudioFilter(asdr_batch_t *b, int ch);
void asdr_disableAudioFilter(asdr_batch_t *b, int ch);
void asdr_setAudioFilter(asdr_batch_t *b, int ch, int filter);
int asdr_getAudioFilter(asdr_batch_t *b, int ch);

/* ---- ALS notch/peak filter (AudioSDR.h:107-117; AudioSDR.cpp:356-398) */
void asdr_enableALSfilter(asdr_batch_t *b, int ch);
void asdr_disableALSfilter(asdr_batch_t *b, int ch);
void asdr_setALSfilterNotch(asdr_batch_t *b, int ch);
void asdr_setALSfilterPeak(asdr_batch_t *b, int ch);
void asdr_setALSfilterAdaptive(asdr_batch_t *b, int ch);
void asdr_setALSfilterStatic(asdr_batch_t *b, int ch);
void asdr_setALSfilterParams(asdr_batch_t *b, int ch, unsigned int m, float lambda, float delay);
int asdr_ALSfilterIsEnabled(asdr_batch_t *b, int ch);
int asdr_ALSfilterIsNotch(asdr_batch_t *b, int ch);
int asdr_ALSfilterIsPeak(asdr_batch_t *b, int ch);
int asdr_ALSfilterIsAdaptive(asdr_batch_t *b, int ch);

/* ---- AGC (AudioSDR.h:120-144; AudioSDR.cpp:495-600) */
void asdr_enableAGC(asdr_batch_t *b, int ch);
void asdr_disableAGC(asdr_batch_t *b, int ch);
int asdr_AGCisEnabled(asdr_batch_t *b, int ch);
int asdr_AGCisActive(asdr_batch_t *b, int ch);               /* reads device state */
void asdr_setAGCthreshold(asdr_batch_t *b, int ch, float db);
void asdr_setAGCslope(asdr_batch_t *b, int ch, float slope);
void asdr_setAGCmode(asdr_batch_t *b, int ch, int mode);
void asdr_setAGCkneeWidth(asdr_batch_t *b, int ch, float db);
void asdr_setAGCattackTime(asdr_batch_t *b, int ch, float ms);
void asdr_setAGCreleaseTime(asdr_batch_t *b, int ch, float ms);
void asdr_setAGChangTime(asdr_batch_t *b, int ch, float ms);
void asdr_setAGCstaticGain(asdr_batch_t *b, int ch, float gain);
float asdr_getAGCthreshold(asdr_batch_t *b, int ch);
float asdr_getAGCslope(asdr_batch_t *b, int ch);
float asdr_getAGCkneeWidth(asdr_batch_t *b, int ch);
float asdr_getAGCattack(asdr_batch_t *b, int ch);
float asdr_getAGCrelease(asdr_batch_t *b, int ch);
float asdr_getAAGalphaAttack(asdr_batch_t *b, int ch);       /* sic: AudioSDR.h:137 */
float asdr_getAGCbetaAttack(asdr_batch_t *b, int ch);
float asdr_getAGCalphaRelease(asdr_batch_t *b, int ch);
float asdr_getAGCbetaRelease(asdr_batch_t *b, int ch);
float asdr_getAGClookup(asdr_batch_t *b, int ch, int i);
float asdr_getAGCstaticGain(asdr_batch_t *b, int ch);
float asdr_getAMcarrierLevel(asdr_batch_t *b, int ch);       /* reads device state */
/* getAGCmakeUpMode (AudioSDR.h:144) is declared but never defined upstream: not exported. */

/* ---- impulse noise blanker (AudioSDR.h:147-152; AudioSDR.cpp:653-682) */
void asdr_enableNoiseBlanker(asdr_batch_t *b, int ch);
void asdr_disableNoiseBlanker(asdr_batch_t *b, int ch);
void asdr_setNoiseBlankerThreshold(asdr_batch_t *b, int ch, float ratio);
void asdr_setNoiseBlankerThresholdDb(asdr_batch_t *b, int ch, float db);
int asdr_NoiseBlankerisEnabled(asdr_batch_t *b, int ch);
int asdr_NoiseBlankerDetection(asdr_batch_t *b, int ch);     /* reads device state */

/* ---- synchronous AM detector (AudioSDR.h:155-156; AudioSDR.cpp:752-757) */
float asdr_getSAMfrequency(asdr_batch_t *b, int ch);         /* reads device state */
int asdr_getSAMphaseLockStatus(asdr_batch_t *b, int ch);     /* reads device state */

/* ---- batch-only additions (no reference analogue) ------------------------------------------- */
/* Bulk status read-back after a batch: one int32/float per channel into caller arrays (any may be NULL). */
int asdr_read_status(asdr_batch_t *b, int32_t *agc_active, int32_t *nb_detected, int32_t *sam_locked,
                     float *sam_frequency, float *am_carrier);
/* Debug taps: when enabled, the next asdr_update*() calls also record the float32 stage outputs of the
 * LAST block processed, as [tap][channel][128]; tap order = ASDR_TAP_*.  Costs extra HBM traffic. */
enum { ASDR_TAP_SCALED_I = 0, ASDR_TAP_SCALED_Q, ASDR_TAP_NB_I, ASDR_TAP_NB_Q, ASDR_TAP_IF_I, ASDR_TAP_IF_Q,
       ASDR_TAP_MIX_I, ASDR_TAP_MIX_Q, ASDR_TAP_DEMOD, ASDR_TAP_AUDIO_FILT, ASDR_TAP_AGC, ASDR_TAP_ALS,
       ASDR_N_TAPS };
int asdr_enable_taps(asdr_batch_t *b, int on);
int asdr_read_taps(asdr_batch_t *b, float *dst /* [ASDR_N_TAPS][n_channels][128] */);
/* Timing (none of it in the reference).  By default a call enqueues its kernels and nothing else: every HIP event record is
 * one more packet between two kernels (a C2 call costs 0.134 ms back to back with an event pair around it, 0.127 ms without).
 *   asdr_set_launch_timing(b, 1)   from now on every call records an event pair around its launches;
 *   asdr_last_kernel_ms            elapsed milliseconds of the most recent such call (synchronises on the stop event;
 *                                  -1 when launch timing is off);
 *   asdr_region_timing_begin/_end  ONE event pair on `stream` around any number of calls: _end records the stop event on the
 *                                  same stream, synchronises, and returns the elapsed milliseconds and the number of update
 *                                  calls in between -- the average time per back-to-back call, gaps included, with no packet
 *                                  added between the kernels (what bench.py reports as kernel_ms). */
int asdr_set_launch_timing(asdr_batch_t *b, int on);
float asdr_last_kernel_ms(asdr_batch_t *b);
int asdr_region_timing_begin(asdr_batch_t *b, void *stream);
int asdr_region_timing_end(asdr_batch_t *b, float *ms_total, long *n_calls);
/* Per-launch kernel timing across a region: _begin() arms up to `max_launches` HIP-event pairs; every following
 * asdr_update_device() records one pair on ITS launch stream around the kernel; _end() synchronises, writes the
 * elapsed milliseconds of each launch into ms[0..return-1] and disarms. */
int asdr_kernel_timing_begin(asdr_batch_t *b, int max_launches);
int asdr_kernel_timing_end(asdr_batch_t *b, float *ms, int cap);
/* Control-plane introspection.  A setter only marks the channels it touched; the next update() refills and uploads those
 * parameter rows and rebuilds the wave schedule only if a touched channel's schedule key (kernel instantiation, mode, enables,
 * tables) changed.  asdr_control_plane_flush() runs the HOST half of that step on a control-plane-only batch (ASDR_NO_DEVICE;
 * refused on a device batch, where the next update() does it) and reports what it did:
 *   stats[0] = parameter rows refilled, stats[1] = 1 if the schedule was rebuilt, stats[2] = waves in the plain / SAM / ALS
 *   sub-ranges (ALS: the three kinds with the filter enabled together) packed as plain | sam << 21 | als << 42, stats[3] = AGC gain tables alive in the pool. */
int asdr_control_plane_flush(asdr_batch_t *b, long long stats[4]);
/* The derived constants the hot path runs with, as the host evaluated them from the reference's in-class initialisers and
 * setters (AudioSDR.h:238-239, 249-284; AudioSDR.cpp:447, 563-566) -- exposed so that an independent restatement of that
 * arithmetic can pin them bit for bit (tests/test_control_plane_independent.py).  out[12] = pll_b0, pll_b1, pll_a1, alpha_freq,
 * beta_freq, f_conv, lock_freq_low, lock_freq_high, twoPI, halfPI, twoPI/AUDIO_SAMPLE_RATE_EXACT, 1 - nb alpha; returns the
 * AGC hang count (samples) of channel `ch` (0 for a bad channel). */
unsigned int asdr_get_chain_constants(asdr_batch_t *b, int ch, float out[12]);
/* Mode values outside 0..6 (reachable through setDemodMode, AudioSDR.cpp:188).  The reference then runs neither demodulator
 * (AudioSDR.cpp:84, 122) and its audio filter / AGC / ALS / output stage process the member _audioOut AGAIN, i.e. the audio the
 * previous block left there (:149-161).  To reproduce that, every block of every channel stores its post-ALS float audio row
 * (512 B per channel-block written; the kernel cannot know which block is a channel's last with a known mode).  This is the
 * DEFAULT (on = 1): results equal the reference's for every mode value.  on = 0 drops the row: unknown mode values then process
 * a silent block (everything else unchanged) and the 512 B per channel-block are saved -- for callers that never set one.
 * Switching it back on starts from silent rows. */
int asdr_set_exact_unknown_mode(asdr_batch_t *b, int on);
int asdr_get_exact_unknown_mode(asdr_batch_t *b);
/* How many calls so far ran as the streaming block pipeline (a multi-block call on a small batch of SSB-class or AM channels:
 * three role-specialised waves per group of 8 channels -- blanker + IF | mixer + Hilbert | audio filter + AGC + output -- work
 * on consecutive blocks at the same time).  Results are bit-identical to the block-by-block path; the counter
 * exists so that tests and benchmarks can tell which path ran. */
long asdr_stream_pipeline_launches(asdr_batch_t *b);
/* The pipeline is a transaction.  Its roles wait for each other with BOUNDED waits, so all 3 w + 1 workgroups must be resident at
 * the same time; asdr_create() asks the runtime how many the device holds (occupancy x compute units, at most one per compute
 * unit) and batches with more channel groups than asdr_stream_pipeline_max_groups() never take the pipeline.  Should a wait still
 * run out (the GPU shared with another long kernel, masked compute units), every wave of the launch leaves, and the launches
 * enqueued behind it on the same stream put the channels' state back from a snapshot taken in front of the pipeline and run the
 * call again on the in-kernel block loop: the caller's next operation on that stream sees exact results and exact state either
 * way, without any host synchronisation.  asdr_stream_pipeline_recoveries() = how many calls went that way (synchronises).
 * asdr_set_stream_pipeline(b, 0) opts a batch out of the pipeline altogether (default on; environment ASDR_NO_STREAM_PIPELINE=1
 * at asdr_create time = default off). */
long asdr_stream_pipeline_recoveries(asdr_batch_t *b);
/* Round 6: the pipeline's role-2 workgroup shares the Hilbert FIR between its role wave and ONE helper wave (asdr_stream_kernel, 128 threads per
 * workgroup, six of them per compute unit) or THREE (asdr_stream_kernel_h3, 256 threads: the FIR in quarters).  The three-helper form is taken
 * while every pipeline workgroup has a compute unit to itself (3 x channel groups <= compute units: the bank sizes the pipeline serves best,
 * where three quarters of the chip idle) and the residency arithmetic holds for it.  asdr_set_stream_fir_helpers(b, -1 | 0 | 1): by that rule
 * (default) | always one helper | three wherever they are resident (environment ASDR_STREAM_H3=0|1 at asdr_create time);
 * asdr_stream_pipeline_h3_calls() = calls that took the three-helper form.  Results are bit-identical either way. */
int asdr_set_stream_fir_helpers(asdr_batch_t *b, int mode);
long asdr_stream_pipeline_h3_calls(asdr_batch_t *b);
/* The pipeline's exchange rings, progress counters and snapshot (about 9 KB per channel) are allocated at its first use.  If that
 * allocation fails the batch opts itself out of the pipeline and keeps the in-kernel block loop -- the call still succeeds; this
 * counts such events.  A call whose output rows overlap its input rows (in-place use, the reference's own convention:
 * AudioSDR.cpp:158-165 writes the audio into blockI) never takes the pipeline either: its recovery restores channel state, not
 * caller buffers. */
long asdr_stream_pipeline_alloc_failures(asdr_batch_t *b);
/* A pipeline call that carries sub-ranges BESIDE the pipeline (remainders, a few SAM / ALS channels: long-running kernels on helper
 * streams) keeps resident-workgroup headroom for them: 3 x groups + side waves + a margin must fit the device's occupancy answer, or the
 * call takes the other launch forms (results identical; this counts such calls).  Without side sub-ranges the pipeline may fill the
 * device.  Streams are shared by all batches of a process on a device (one pool of three: see asdr_order_after), so another batch's
 * kernels can still delay a pipeline's role waves -- then the bounded waits fire and the call is re-run from its snapshot
 * (asdr_stream_pipeline_recoveries): exact, slower. */
long asdr_stream_pipeline_headroom_refusals(asdr_batch_t *b);
int asdr_stream_pipeline_max_groups(asdr_batch_t *b);
int asdr_set_stream_pipeline(asdr_batch_t *b, int on);
/* Large sub-ranges of the schedule as SEVERAL kernels on as many streams (bit-identical results: the pieces touch disjoint channels).
 * Kernels of one stream run one after the other and each drains its last waves before the next may start; pieces on different streams
 * fill each other's tails.  pieces = 1..8 (1 = one kernel per sub-range), min_waves = the smallest sub-range (in waves of 8 channels)
 * that is cut (<= 0: keep the current value, default 2048).  Default at asdr_create time: ASDR_LAUNCH_SPLIT_DEFAULT, or the environment
 * (ASDR_LAUNCH_SPLIT, ASDR_LAUNCH_SPLIT_MIN_WAVES). */
int asdr_set_launch_split(asdr_batch_t *b, int pieces, int min_waves);
/* How a batch launches its SAM channels (bit-identical results): fused = 1 -> the fused 4-wave kernel always; otherwise the three
 * launches pre | PLL | post from split_min_channels SAM channels on (<= 0: the default, 512).  Defaults at asdr_create time from
 * the environment (ASDR_SAM_FUSED, ASDR_SAM_SPLIT_MIN): comparison switches of the measurement tools. */
int asdr_set_sam_launch_form(asdr_batch_t *b, int fused, int split_min_channels);
/* Channels with a short ALS filter (taps <= 64, delay + taps <= 65; not SAM) CAN run as two launches -- the chain up to the AGC as the
 * plain kernel, then the filter and the output stage on small LDS rows (asdr_als_kernel) -- when the batch has at least
 * split_min_channels of them.  Default (<= 0, or environment ASDR_ALS_SPLIT_MIN unset at asdr_create time): never -- measured on
 * MI355X the two-launch form is 2 % slower for 131,072 mixed channels, 12 % slower for 1,048,576 and 9 % slower for an all-ALS batch
 * (profiles/README.md; it is the measuring bench that found the filter's LDS bound).  Bit-identical results.  asdr_schedule_layout()'s out[7] carries the choice in bit 1. */
int asdr_set_als_launch_form(asdr_batch_t *b, int split_min_channels);
/* Test hook: the number of polls after which a pipeline wait gives up (0 = the default, 2^18).  A tiny value injects timeouts, so
 * that the recovery path can be tested on an idle GPU. */
int asdr_debug_set_stream_spin_limit(asdr_batch_t *b, unsigned int polls);
/* Test / experiment hook: lowers the largest number of channel groups the pipeline takes (asdr_stream_pipeline_max_groups();
 * default = what the occupancy query of asdr_create says the device holds at once: 512 on an MI355X).  Values beyond the query's
 * answer are clamped to it -- more groups than resident workgroups cannot make progress and every call would sit out its 2^18
 * polls before the recovery launches run. */
int asdr_debug_set_stream_max_groups(asdr_batch_t *b, int groups);
/* The wave schedule as the last flush built it (the next update's launches; a control-plane-only batch: after
 * asdr_control_plane_flush), in schedule slots (8 per wave): out[0..4] = the sub-ranges of whole waves of one settings group by
 * kernel kind -- 0 plain, 1 SAM, 2 ALS on the long rows, 3 ALS on the compact rows (taps <= 64, delay + taps <= 65, not SAM),
 * 4 SAM + such an ALS filter -- (kind 1 also holds the SAM remainders when a batch with fewer than 512 SAM channels runs them in
 * the fused SAM kernel), out[5] = the sub-range of all other remainders (< 8 channels per settings group), out[6] = the kind of the
 * general kernel that runs it (0 or 2; -1 if empty), out[7] bit 0 = SAM channels run as pre | PLL | post launches, bit 1 = short ALS filters run as chain | filter launches.  For tests and
 * capacity planning; DESIGN.md 3.1. */
int asdr_schedule_layout(asdr_batch_t *b, int out[8]);
/* Library / build identification string (contains "gfx950"). */
const char *asdr_version(void);
/* Launch census (process-wide): how often each kernel of the chain has been launched since the last reset -- so that a measurement can name the
 * instantiation it timed (bench.py `roofline.kernel`, `kernels_launched`).  asdr_kernels_count() names, 0 <= i < count. */
int asdr_kernels_count(void);
const char *asdr_kernels_name(int i);
unsigned long long asdr_kernels_launches(int i);
void asdr_kernels_launches_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* ASDR_H_ */
