/* receiver_bank.c -- a plain C99 host over the C ABI of libasdr_hip.so (include/asdr.h): TEST PROGRAM.
 *
 * What a C maintainer of the reference would write to replace N `AudioSDR` objects (AudioSDR.h:75-156) by one batch:
 *   - a SHARDED batch (asdr_create_sharded; here two shards on one device), configured through the reference's method names with
 *     GLOBAL channel indices: every receiver as EXTRAS/BareBonesWSPR/BareBonesWSPR.ino:87-102 sets its AudioSDR up, a few of them in
 *     other modes;
 *   - host-resident 128-sample blocks in page-locked memory (asdr_host_alloc), one asdr_update() per audio period -- the reference's
 *     own data path (AudioSDR.cpp:46-47, 158-167);
 *   - status getters by global index.
 * The program links the CPU oracle (oracle/libasdr_oracle.so: allowed here, this is tests/) and compares every output sample and
 * the getters of every receiver with N oracle instances fed the same blocks.  Exit code 0 = bit-exact.
 *   usage: receiver_bank <device> [n_receivers] [n_periods] [blocks_per_period] [pinned: 1 | 0 = ordinary malloc'ed rows]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "asdr.h"
#include "asdr_oracle.h"

#define PI_D 3.14159265358979323846

static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s; }

/* tone + noise, int16, one receiver-block (not the Python generator: any deterministic input does -- both sides get the same) */
static void make_block(int16_t *I, int16_t *Q, int rx, long t0, uint32_t *seed) {
  const double fc = 1500.0 + 1.4648 * (rx % 4) + 0.7 * (rx % 13), A = 0.02 + 0.2 * ((rx % 5) / 5.0);
  for (int k = 0; k < ASDR_BLOCK_SAMPLES; k++) {
    const double ph = 2.0 * PI_D * fc * (double)(t0 + k) / 44100.0;
    const double ni = ((double)(lcg(seed) >> 8) / 16777216.0 * 2.0 - 1.0) * 0.05, nq = ((double)(lcg(seed) >> 8) / 16777216.0 * 2.0 - 1.0) * 0.05;
    I[k] = (int16_t)(32767.0 * (A * cos(ph) + ni));
    Q[k] = (int16_t)(32767.0 * (A * sin(ph) + nq));
  }
}

int main(int argc, char **argv) {
  const int device = argc > 1 ? atoi(argv[1]) : 0;
  const int n = argc > 2 ? atoi(argv[2]) : 200, periods = argc > 3 ? atoi(argv[3]) : 6, T = argc > 4 ? atoi(argv[4]) : 3;
  int devs[2] = {device, device};
  asdr_batch_t *rx = asdr_create_sharded(n, 2, devs);
  if (!rx) { fprintf(stderr, "asdr_create_sharded: %s\n", asdr_last_error()); return 2; }
  asdr_oracle_t **ref = (asdr_oracle_t **)malloc(sizeof(*ref) * (size_t)n);
  for (int c = 0; c < n; c++) ref[c] = ao_create();

  /* BareBonesWSPR.ino:87-102, 129 -- once for the whole bank ... */
  asdr_enableAGC(rx, ASDR_ALL); asdr_setAGCmode(rx, ASDR_ALL, ASDR_AGCmedium); asdr_disableALSfilter(rx, ASDR_ALL);
  asdr_disableNoiseBlanker(rx, ASDR_ALL); asdr_setNoiseBlankerThresholdDb(rx, ASDR_ALL, 10.0f);
  asdr_setInputGain(rx, ASDR_ALL, 1.0f); asdr_setOutputGain(rx, ASDR_ALL, 0.5f); asdr_setIQgainBalance(rx, ASDR_ALL, 1.020f);
  asdr_setAudioFilter(rx, ASDR_ALL, ASDR_audioWSPR);
  const float offset = asdr_setDemodMode(rx, ASDR_ALL, ASDR_WSPRmode);
  asdr_setMute(rx, ASDR_ALL, 0);
  for (int c = 0; c < n; c++) {
    asdr_oracle_t *o = ref[c];
    ao_enableAGC(o); ao_setAGCmode(o, ASDR_AGCmedium); ao_disableALSfilter(o); ao_disableNoiseBlanker(o); ao_setNoiseBlankerThresholdDb(o, 10.0f);
    ao_setInputGain(o, 1.0f); ao_setOutputGain(o, 0.5f); ao_setIQgainBalance(o, 1.020f); ao_setAudioFilter(o, ASDR_audioWSPR);
    if (ao_setDemodMode(o, ASDR_WSPRmode) != offset) { fprintf(stderr, "tuning offset differs\n"); return 1; }
    ao_setMute(o, 0);
  }
  /* ... and a few receivers by global index, on both sides of the shard boundary (n / 2) */
  const int odd[4] = {3, n / 2 - 1, n / 2, n - 2};
  const int odd_mode[4] = {ASDR_USBmode, ASDR_AMmode, ASDR_SAMmode, ASDR_CW_LSBmode};
  for (int k = 0; k < 4; k++) {
    asdr_setDemodMode(rx, odd[k], odd_mode[k]); asdr_enableNoiseBlanker(rx, odd[k]); asdr_enableAudioFilter(rx, odd[k]);
    ao_setDemodMode(ref[odd[k]], odd_mode[k]); ao_enableNoiseBlanker(ref[odd[k]]); ao_enableAudioFilter(ref[odd[k]]);
  }

  const size_t row = (size_t)T * ASDR_BLOCK_SAMPLES, bytes = (size_t)n * row * sizeof(int16_t);
  int pinned = argc > 5 ? atoi(argv[5]) : 1;
  int16_t *I = NULL, *Q = NULL, *out = NULL;
  if (pinned) { I = (int16_t *)asdr_host_alloc(bytes); Q = (int16_t *)asdr_host_alloc(bytes); out = (int16_t *)asdr_host_alloc(bytes); }
  if (!I || !Q || !out) {   /* ordinary memory: the library stages it (also where page-locking is not to be had) */
    if (I) asdr_host_free(I);
    if (Q) asdr_host_free(Q);
    if (out) asdr_host_free(out);
    pinned = 0;
    I = (int16_t *)malloc(bytes); Q = (int16_t *)malloc(bytes); out = (int16_t *)malloc(bytes);
  }
  int16_t want[ASDR_BLOCK_SAMPLES];
  if (!I || !Q || !out) { fprintf(stderr, "out of memory\n"); return 2; }
  uint32_t *seed = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
  for (int c = 0; c < n; c++) seed[c] = 12345u + (uint32_t)c;
  long bad = 0, t = 0;
  for (int p = 0; p < periods; p++) {
    for (int c = 0; c < n; c++)
      for (int b = 0; b < T; b++) make_block(I + c * row + (size_t)b * ASDR_BLOCK_SAMPLES, Q + c * row + (size_t)b * ASDR_BLOCK_SAMPLES, c, t + (long)b * ASDR_BLOCK_SAMPLES, &seed[c]);
    if (asdr_update(rx, I, Q, out, T) != 0) { fprintf(stderr, "asdr_update: %s\n", asdr_last_error()); return 2; }
    for (int c = 0; c < n; c++)
      for (int b = 0; b < T; b++) {
        const size_t o = c * row + (size_t)b * ASDR_BLOCK_SAMPLES;
        ao_update(ref[c], I + o, Q + o, want);
        for (int k = 0; k < ASDR_BLOCK_SAMPLES; k++) bad += (out[o + k] != want[k]);
      }
    if (p == periods / 2) {   /* a setter between two audio periods, by global index */
      asdr_setAGCstaticGain(rx, n / 2 + 1, 4.0f); ao_setAGCstaticGain(ref[n / 2 + 1], 4.0f);
    }
    t += (long)T * ASDR_BLOCK_SAMPLES;
  }
  int info[2] = {0, 0};
  asdr_host_path_info(rx, info);
  long bad_status = 0;
  for (int c = 0; c < n; c++) {
    bad_status += asdr_AGCisActive(rx, c) != ao_AGCisActive(ref[c]);
    bad_status += asdr_getSAMphaseLockStatus(rx, c) != ao_getSAMphaseLockStatus(ref[c]);
    const float a = asdr_getAMcarrierLevel(rx, c), b2 = ao_getAMcarrierLevel(ref[c]);
    bad_status += memcmp(&a, &b2, sizeof a) != 0;
    const float f1 = asdr_getSAMfrequency(rx, c), f2 = ao_getSAMfrequency(ref[c]);
    bad_status += memcmp(&f1, &f2, sizeof f1) != 0;
  }
  printf("receivers %d shards %d periods %d blocks %d pinned %d samples_differ %ld status_differ %ld offset %.1f\n", n, asdr_n_shards(rx), periods, T,
         info[1], bad, bad_status, offset);
  if (pinned) { asdr_host_free(I); asdr_host_free(Q); asdr_host_free(out); } else { free(I); free(Q); free(out); }
  for (int c = 0; c < n; c++) ao_destroy(ref[c]);
  free(ref); free(seed);
  asdr_destroy(rx);
  return (bad == 0 && bad_status == 0) ? 0 : 1;
}
