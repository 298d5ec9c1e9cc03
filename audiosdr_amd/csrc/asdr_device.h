/* asdr_device.h -- data layout shared by the host control plane (asdr_host.cpp) and the HIP
 * kernels (asdr_kernels.hip).  Everything here is plain-old-data living in HBM.
 *
 * HBM layout (one row per channel in every array; rows of bulk history are 512 B multiples so
 * that a 64-lane wave reading "8 lanes x 16 floats" per channel issues 64-B-per-lane dwordx4
 * loads, fully coalesced within each row):
 *
 *   params   ChanParams  [C+1]            96 B  read-only per launch (row C = dummy channel)
 *   small    ChanSmall   [C+1]           448 B  biquad states, phases, AGC/PLL/NB scalars, ring slots
 *   nb_hist  int16       [C+1][3][2][128] 1.5 KiB noise-blanker ring of RAW samples: slot x {I,Q} x sample (oldest, middle,
 *                                                 newest); the gains in force at arrival are in ChanSmall.nb_gain
 *   nb_mask  uint8       [C+1][160]        160 B  mask codes of (middle block + 10 look-ahead), 138 used
 *   hil_q    float       [C+1][2][128]     1 KiB  Hilbert Q history ring (2 previous shifted blocks)
 *   hil_i    float       [C+1][2][128]     1 KiB  ring of mixed I blocks (this block, previous = the 128-sample delay)
 *   als_x    float       [C+1][2][128]     1 KiB  ring of ALS input blocks (this block, previous block); position = UpdateArgs.als_phase
 *   als_w    float       [C+1][128]        512 B  ALS coefficients
 *   audio_prev float     [C+1][128]        512 B  _audioOut as the block left it (after the audio filter / AGC / ALS): what an
 *                                                 unknown mode value re-processes (AudioSDR.cpp:84,122,149-161)
 *   agc_tab  float       [T][132]                 pool of distinct AGC gain tables (130 used)
 */
#ifndef ASDR_DEVICE_H_
#define ASDR_DEVICE_H_

#include <stdint.h>

#define ASDR_N 128
#define ASDR_PLL_WRAP_MAX 64  /* turns of 2*pi the SAM PLL's phase wrap may take per sample before the estimate is reset */

/* ChanParams.flags */
#define ASDR_F_NB_EN        (1u << 0)
#define ASDR_F_AF_EN        (1u << 1)
#define ASDR_F_AGC_EN       (1u << 2)
#define ASDR_F_ALS_EN       (1u << 3)
#define ASDR_F_ALS_NOTCH    (1u << 4)
#define ASDR_F_ALS_ADAPTIVE (1u << 5)
#define ASDR_F_MUTED        (1u << 6)

/* ChanParams.reset: state to re-initialise before the next block (set by setters, consumed by
 * the reset kernel) */
#define ASDR_R_IF   (1u << 0) /* IF biquad state -> 0        (setDemodMode, AudioSDR.cpp:191-218) */
#define ASDR_R_AF   (1u << 1) /* audio biquad state -> 0     (setAudioFilter, AudioSDR.cpp:300-309) */
#define ASDR_R_NB   (1u << 2) /* NB buffers 0, mask 1        (initBlanker, AudioSDR.cpp:676-682) */
#define ASDR_R_ALS  (1u << 3) /* ALS taps + history -> 0     (enableALSfilter, AudioSDR.cpp:384-391) */
#define ASDR_R_IMG  (1u << 4) /* AM image biquad state -> 0  (init, AudioSDR.cpp:178-179) */
#define ASDR_R_ALL  (1u << 5) /* everything to power-on state (asdr_create) */

typedef struct {
  uint32_t mode;        /* _mode (uint16 in the reference) */
  uint32_t flags;
  uint32_t reset;
  int32_t if_table;     /* row of asdr_bq_pool for the IF band-pass */
  int32_t audio_table;  /* row of asdr_bq_pool for the audio filter */
  int32_t agc_table;    /* row of the AGC table pool */
  float in_gain_i, in_gain_q;
  float output_gain;
  float freq_shift;
  float nb_threshold;
  float agc_alpha_att, agc_beta_att, agc_alpha_rel, agc_beta_rel, agc_static_gain;
  uint32_t agc_hang_count;
  int32_t als_m, als_delay;
  float als_lambda;
  uint32_t pad_[4];
} ChanParams; /* 24 words = 96 B */

typedef struct {
  float if_state[2][16];  /* [I,Q][stage][x1,x2,y1,y2]   _IFfilterStateI/Q   AudioSDR.h:191-192 */
  float img_state[2][16]; /*                             _AMimage_stateI/Q  AudioSDR.h:193-194 */
  float af_state[16];     /*                             _audio_filter_state AudioSDR.h:195     */
  float phase_ssb, phase_am;          /* AudioSDR.cpp:43-44 */
  float nb_avg;                       /* _nb_AvgMag */
  float am_carrier;                   /* _agc_AMcarrierLevel */
  float agc_gain, agc_old_abs;        /* _agc_gain, _old_absVal */
  uint32_t agc_hang_counter;
  float pll_y_re, pll_y_im, pll_prev_filt;      /* AudioSDR.cpp:690-692 statics */
  float pll_d0, pll_d1, pll_phase_est, pll_freq; /* delay0, delay1, phase_est, _PLLfreq */
  uint32_t nb_slot_unused, hil_slot;  /* hil_slot: Hilbert ring parity (the blanker ring position is batch-wide: UpdateArgs.nb_phase) */
  uint32_t status;                    /* ASDR_S_* bits */
  float nb_gain[3][2];                /* input gains {I,Q} in force when each noise-blanker ring slot was written */
  uint32_t pad_[9];
} ChanSmall; /* 80 + 17 + 15 = 112 words = 448 B */

#define ASDR_S_AGC_ACTIVE  (1u << 0) /* _agc_is_active (in-class initialiser: true) */
#define ASDR_S_NB_DETECTED (1u << 1) /* _nb_impulseDetected */
#define ASDR_S_PLL_LOCKED  (1u << 2) /* _SAM_PLL_isLocked */

#define ASDR_NB_MASK_ROW 160
#define ASDR_NB_MASK_USED 138
#define ASDR_AGC_TAB_ROW 132

/* constants identical for every channel, evaluated once on the host exactly as the reference's
 * in-class initialisers evaluate them (AudioSDR.h:164-168, 238-239, 249-284) */
typedef struct {
  double inv_two_pi_d;          /* RN(1.0 / (double)two_pi_f): reciprocal for the exact sine-index division */
  double sin_index_scale_d;     /* RN(RN(65535.0 / (double)two_pi_f) * (1 + 2^-49)): sin_f32's table phase (long)(Phase * 65535.0 / twoPI)
                                   (AudioSDR.h:364) as ONE binary64 multiply + truncation (asdr_kernels.hip sin_index; proof and exhaustive
                                   check: oracle ao_check_sin_index_one_multiply) */
  double half_pi_d;             /* PI / 2.0 (Arduino PI, double): cos_f32's phase offset, AudioSDR.h:376 -- a kernel argument so that it
                                   lives in SGPRs (as a literal the compiler parks it in a VGPR pair for the whole kernel) */
  float if_center;              /* 6890.0f */
  float two_pi_f;               /* (float)(2.0*PI) */
  float half_pi_f;              /* (float)(0.5*PI) */
  float phase_inc_unit;         /* two_pi_f / 44100.0f : freq_shifter's twoPI/AUDIO_SAMPLE_RATE_EXACT */
  float nb_alpha, nb_beta;
  float pll_b0, pll_b1, pll_a1;
  float pll_alpha_freq, pll_beta_freq, pll_f_conv, pll_lock_lo, pll_lock_hi;
} ChainConsts;

typedef struct {
  int32_t ch;
  uint32_t mode, flags;
  uint32_t lo;   /* local-oscillator cache: low byte = 1 + the cache entry of the slot's settings group (0 = none), bit 8 = the slot's wave
                    is the group's first one and leaves the NEXT block's pairs in that entry (ASDR_LO_WRITER) */
} SlotInfo;
#define ASDR_LO_ENTRIES 8        /* settings groups with an entry of their own (round 2: one entry, wave 0's group) */
#define ASDR_LO_WRITER 0x100u
#define ASDR_LO_WRITER_LANE(l) (0x100u << (l))   /* lane l >= 1 (asdr_host.cpp, lanes: the pieces of a batch on never-joined streams): the group's
                                                     first wave inside lane l's piece of its sub-range -- the writer of lane l's own set of entries
                                                     (lane 0's writer is the group's first wave: ASDR_LO_WRITER); up to 8 lanes, bits 8..15 */

/* Local-oscillator cache: the mixer's phase sequence and its sin/cos pairs depend only on (carried phase, increment).  When all
 * channels of wave 0 share one pair (receivers configured together), that wave also computes the NEXT block's 128 pairs and
 * leaves them here; a wave of the next launch whose channels carry exactly that phase and increment (compared bit for bit)
 * reads the pairs instead of running the 128-step recurrence and the table lookups itself.  Two entries: a launch reads
 * one and (wave 0) writes the other.  A miss just takes the per-wave path. */
typedef struct {
  uint32_t key_phase, key_inc;   /* bit patterns of the phase at the start of the block and of the increment */
  float phase_end;               /* phase after the block's 128 samples */
  uint32_t pad_;
  float c[ASDR_N], s[ASDR_N];    /* cos_f32 / sin_f32 of the block's 128 phases (AudioSDR.h:358-377) */
} LoEntry;

typedef struct {
  const ChanParams *params;
  ChanSmall *small;
  int16_t *nb_hist;
  uint8_t *nb_mask;
  float *hil_q, *hil_i, *als_x, *als_w;
  float *audio_prev;      /* [n_channels + 1][128] the member _audioOut (AudioSDR.h:169) as the previous block left it, i.e. after the audio
                             filter, AGC and ALS: a mode value outside 0..6 runs neither demodulator (AudioSDR.cpp:84, 122) and the post
                             stages process this row AGAIN (:149-161).  Written by every block of every channel (the kernel cannot know
                             which block is a channel's last with a known mode); NULL = not kept, unknown modes process silence
                             (asdr_set_exact_unknown_mode(b, 0)) */
  const float *agc_tab;
  const SlotInfo *sched;  /* per wave slot: channel index (padded to a multiple of 8 with the dummy channel n_channels),
                             mode and flags -- one 16-B load instead of the dependent chain sched -> params */
  int32_t n_sched;        /* multiple of 8 */
  int32_t n_channels;
  const int16_t *in_i, *in_q;
  int16_t *out;
  int32_t n_blocks;       /* blocks per channel processed by this call */
  uint32_t nb_phase;      /* blocks processed by the batch so far, mod 3: oldest slot of every channel's blanker ring */
  uint32_t als_phase;     /* blocks processed by the batch so far, mod 2: the slot of the ALS input ring (als_x) THIS block's input goes to;
                             the other slot holds the previous block's (enableALSfilter zeroes both, AudioSDR.cpp:384-391, so a channel
                             whose filter was off meanwhile never sees a stale position) */
  LoEntry *lo_cache;      /* [2][ASDR_LO_ENTRIES] */
  uint32_t lo_parity;     /* half this launch reads; a settings group's first wave (SlotInfo.lo) fills its entry of the other half */
  uint32_t lo_write;      /* the launch may fill entries (off for the pipeline, whose oscillator role serves its own ring) */
  uint32_t lo_writer_bit; /* which bit of SlotInfo.lo names this launch's writers: ASDR_LO_WRITER, or ASDR_LO_WRITER_LANE(l) in lane l's launches */
  uint32_t direct_lo;     /* SlotInfo.lo of a direct_ch0 launch's single settings group (writer = its wave 0) */
  int32_t direct_ch0;     /* >= 0: the launched sub-range is ONE key group of consecutive channels direct_ch0, direct_ch0 + 1, ...:
                             slot i is channel direct_ch0 + i with direct_mode / direct_flags, and no wave has to load its
                             schedule entry first (one HBM round trip less at the start of every wave); -1: read a.sched */
  uint32_t direct_mode, direct_flags;
  int32_t in_stride;      /* row stride of I and Q in blocks (>= n_blocks; == n_blocks for the packed layout) */
  int32_t out_stride;     /* row stride of out in blocks (a capture row holds many calls' worth) */
  float *taps;            /* NULL or [ASDR_N_TAPS][n_channels][128] */
  /* block pipeline of the streaming kernel (asdr_stream_kernel): three role-specialised waves per group of 8 channels
   * (blanker + IF | mixer + Hilbert | audio filter + AGC + output) work on consecutive blocks of one call at the same time */
  float *xch_a;           /* [n_channels][ASDR_STREAM_DEPTH][2][128]: IF output I, Q of the blocks in flight (role 1 -> role 2) */
  float *xch_b;           /* [n_channels][ASDR_STREAM_DEPTH][128]:    demodulated audio (role 2 -> role 3) */
  float *xch_c;           /* [n_channels][ASDR_STREAM_DEPTH]:         AM carrier level after the block (role 2 -> role 3's AGC, AudioSDR.cpp:141, 407) */
  uint32_t *stream_prog;  /* [3][stream_waves] blocks completed by each role's wave in this launch (zeroed before it); [3 * stream_waves] =
                             the oscillator role's progress */
  uint32_t *stream_err;   /* [0] set to 1 by a wave that gave up waiting (stream_spin_limit polls): every wave of the launch then leaves
                             at its next wait, and the launches the host enqueues BEHIND the pipeline (asdr_stream_restore_kernel, the
                             gated in-kernel block loop, asdr_stream_ack_kernel) put the channels' state back and run the call again:
                             a pipeline that could not make progress costs time, never results.  [1] = calls recovered that way so far */
  uint32_t stream_spin_limit;   /* polls before a wait gives up (ASDR_STREAM_SPIN_LIMIT; tests shrink it to inject a timeout) */
  const uint32_t *run_if; /* non-NULL: the launch is the pipeline's fallback -- every wave returns at once unless *run_if != 0 */
  int32_t stream_waves;   /* waves per role = (workgroups - 1) / 3 */
  float *xch_sam;         /* SAM sub-range as three launches (pre | PLL | post, asdr_launch_update): the IF output I, Q of the current block,
                             one 8 KB tile per group of 8 schedule slots, [tile][sample][I, Q][slot in tile] -- the pre kernel's wave
                             writes its tile whole, the PLL kernel's lanes (one channel each) read 32-byte segments of 8 tiles;
                             NULL = the fused SAM kernel */
  uint32_t *sam_lock;     /* ... and the PLL's lock flag of the block, one word per schedule slot of the launch, beside the tiles (not
                             through the status word: when the three roles of consecutive blocks overlap -- pre(k+1) | PLL(k+1) beside
                             post(k), asdr_launch_sam_role -- the status word may already hold block k+1's flag when post(k) reads it) */
  uint32_t sam_set, sam_sets;           /* chunked SAM role streams (launches of several blocks): block k of the launch uses tile set (sam_set + k) % sam_sets ... */
  uint32_t sam_set_stride, sam_lock_stride;   /* ... sets sam_set_stride floats (tiles) / sam_lock_stride words (lock flags) apart; all 0 = one set */
  float *als_stage;       /* ALS role streams (a small bank's multi-block call as chain | filter launches on two event-chained streams, asdr_host.cpp):
                             [n_channels][ASDR_ALS_STAGE_SLOTS][128] post-AGC rows of the blocks in flight -- the chain launch of block b stores
                             its row in slot b % S (and in the als_x ring as ever), the filter launch of block b reads slots b % S and (b - 1) % S
                             = the previous block's, so that the chain launches run up to S - 1 blocks ahead of the filter launches without
                             touching what those read; NULL = the als_x ring */
  uint32_t als_stage_cur, als_stage_prev;   /* slots of the launch's first block and of the block in front of it (block k of the launch: + k, mod the slots) */
  LoEntry *lo_ring;       /* [ASDR_LO_RING] the streaming pipeline's oscillator role leaves block b's pairs in entry b % ASDR_LO_RING;
                             its progress counter is stream_prog[3 * stream_waves] */
  ChainConsts k;
} UpdateArgs;
#define ASDR_ALS_STAGE_SLOTS 32  /* ALS role streams: slots of UpdateArgs.als_stage per channel (16 KB) */
#define ASDR_ALS_CHUNK 8         /* ... and blocks per chain / filter launch: the chain launches run up to STAGE_SLOTS / CHUNK - 2 chunks ahead */
#define ASDR_STREAM_DEPTH 4
#define ASDR_LO_RING 8
#define ASDR_STREAM_SPIN_LIMIT (1u << 18)   /* bounded waits: a pipeline whose roles are not co-resident ends with the error flag set and is
                                               re-run on the in-kernel block loop from a snapshot of the state (asdr_host.cpp), never with a hung GPU */
#define ASDR_STREAM_FAIL 0xFFFFFFFFu        /* stream_wait: gave up (or another wave of the launch had) */
/* per-channel state a pipeline call advances (plain instantiation, SSB-class modes): what the snapshot in front of it holds */
#define ASDR_SNAP_BYTES (448 + 1536 + ASDR_NB_MASK_ROW + 1024 + 1024 + 512)   /* ChanSmall, nb_hist, nb_mask, hil_q, hil_i, audio_prev */

/* instantiations of the update kernel (asdr_launch_update) */
#define ASDR_KERNEL_PLAIN 0
#define ASDR_KERNEL_SAM 1
#define ASDR_KERNEL_ALS 2
#define ASDR_KERNEL_ALS_SMALL 3   /* ALS, not SAM, taps <= 64 and delay + taps <= 65: compact rows (asdr_kernels.hip, ALS section) */
#define ASDR_KERNEL_SAM_ALS 4     /* SAM mode + such a short ALS filter: pre | PLL | post launches like _SAM (post with the filter) when the batch
                                     runs its SAM channels that way, the _ALS instantiation otherwise */
#define ASDR_KERNEL_KINDS 5
#ifndef ASDR_SAM_WAVES
#define ASDR_SAM_WAVES 4
#endif   /* waves (x 8 channels) per workgroup of the SAM instantiation: one of them runs every channel's PLL */

#endif /* ASDR_DEVICE_H_ */
