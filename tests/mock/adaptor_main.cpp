// Exercises include/AudioSDR_hip.hpp end to end: sketch-shaped usage (setters once, update() per block).
// argv[1]: HIP device ordinal, or -1 for a control-plane-only check (no GPU).  Prints audio samples for the test.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include "AudioSDR_hip.hpp"
int main(int argc, char **argv) {
  int dev = argc > 1 ? atoi(argv[1]) : 0;
  AudioSDR sdr(dev);
  if (!sdr.ok()) { printf("create failed: %s\n", asdr_last_error()); return 2; }
  float off = sdr.setDemodMode(USBmode);
  sdr.enableAudioFilter(); sdr.setAGCmode(AGCmedium); sdr.setNoiseBlankerThresholdDb(10.0f);
  printf("offset %.1f lower %.1f upper %.1f mode %d agc %d\n", off, sdr.getBPFlower(), sdr.getBPFupper(), (int)sdr.getDemodMode(), (int)sdr.AGCisEnabled());
  if (dev < 0) return 0;
  audio_block_t bi, bq;
  long acc = 0;
  for (int b = 0; b < 8; b++) {
    for (int i = 0; i < 128; i++) { double t = (b * 128 + i) * 2.0 * M_PI * 6290.0 / 44100.0; bi.data[i] = (int16_t)(8000 * cos(t)); bq.data[i] = (int16_t)(8000 * sin(t)); }
    sdr.feed(0, &bi); sdr.feed(1, &bq);
    sdr.update();
    if (sdr.out_[0] != &bi || sdr.out_[1] != &bi) return 3;
    for (int i = 0; i < 128; i++) acc += labs((long)bi.data[i]);
  }
  sdr.feed(0, &bi);            // missing Q: guard must release I and return
  int r0 = sdr.released_; sdr.update();
  printf("released %d guard %d acc %ld\n", sdr.released_, sdr.released_ - r0, acc);
  return acc > 0 ? 0 : 4;
}
