// Exercises include/AudioSDRlib_hip.hpp: the documented graph  source -> AudioIQgenerator -> AudioSDRpreProcessor ->
// {AudioSDR, AudioGrabberComplex256}, one update() per block, sketch-shaped.  argv[1]: HIP device, or -1 for a
// control-plane-only check (no GPU).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "AudioSDRlib_hip.hpp"
int main(int argc, char **argv) {
  int dev = argc > 1 ? atoi(argv[1]) : 0;
  AudioIQgenerator gen(dev);
  AudioSDRpreProcessor pre(dev);
  AudioSDR sdr(dev);
  AudioGrabberComplex256 grab(dev);
  if (!gen.ok() || !pre.ok() || !sdr.ok() || !grab.ok()) { printf("create failed: %s\n", asdr_last_error()); return 2; }
  gen.setGainBalance(1.0f);
  pre.startAutoI2SerrorDetection();
  printf("auto %d corr %d\n", (int)pre.getAutoI2SerrorDetectionStatus(), (int)pre.getI2SerrorCompensation());
  pre.setI2SerrorCompensation(1);
  printf("auto %d corr %d new %d\n", (int)pre.getAutoI2SerrorDetectionStatus(), (int)pre.getI2SerrorCompensation(), (int)grab.newDataAvailable());
  pre.swapIQ(false);
  sdr.setDemodMode(USBmode);
  if (dev < 0) return 0;
  audio_block_t bi, bq;
  int16_t cplx[512];
  long acc = 0; int grabs = 0;
  for (int b = 0; b < 8; b++) {
    for (int i = 0; i < 128; i++) bi.data[i] = (int16_t)(8000 * cos((b * 128 + i) * 2.0 * M_PI * 6290.0 / 44100.0));
    gen.spare_ = &bq; gen.feed(0, &bi); gen.update();                 // real -> I (bi), Q (bq)
    if (gen.out_[0] != &bi || gen.out_[1] != &bq) return 3;
    pre.feed(0, &bi); pre.feed(1, &bq); pre.update();
    if (pre.out_[0] != &bi || pre.out_[1] != &bq) return 4;
    grab.feed(0, &bi); grab.feed(1, &bq); grab.update();
    if (grab.newDataAvailable()) { grab.grab(cplx); grabs++; if (cplx[510] != bi.data[127] || cplx[511] != bq.data[127]) return 6; }
    sdr.feed(0, &bi); sdr.feed(1, &bq); sdr.update();
    for (int i = 0; i < 128; i++) acc += labs((long)bi.data[i]);
  }
  printf("grabs %d acc %ld released %d %d %d\n", grabs, acc, gen.released_, pre.released_, grab.released_);
  return (acc > 0 && grabs == 4) ? 0 : 5;
}
