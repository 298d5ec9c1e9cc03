// Driver of the stand-in build for the blocks either side of the path (tools/ref_shim/README.md): AudioIQgenerator, AudioGrabberComplex256 and
// the paths of AudioSDRpreProcessor that do not need CMSIS's FFT, each the reference's own source compiled by path.
//   ref_front_driver iqgen <balance> <x.bin int16 [blocks][128]> <n_blocks> <out.bin int16 [blocks][2][128]>
//   ref_front_driver grab  <iq.bin int16 [blocks][2][128]> <n_blocks> <grab_after_block> <out.bin int16 [512]>     -> "new <0|1>" on stdout
//   ref_front_driver pre   <correction> <swap> <iq.bin> <n_blocks> <out.bin int16 [blocks][2][128]>                -> getters on stdout
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include "AudioIQgenerator.h"
#include "AudioGrabberComplex256.h"
#include "AudioSDRpreProcessor.h"
SerialShim Serial;
static AudioIQgenerator gen;
static AudioGrabberComplex256 grabber;
static AudioSDRpreProcessor pre;
static std::vector<int16_t> slurp(const char *path, size_t n) { std::vector<int16_t> v(n); FILE *f = fopen(path, "rb"); if (!f || fread(v.data(), 2, n, f) != n) { fprintf(stderr, "cannot read %s\n", path); exit(5); } fclose(f); return v; }
static void dump(const char *path, const std::vector<int16_t> &v) { FILE *f = fopen(path, "wb"); if (!f || fwrite(v.data(), 2, v.size(), f) != v.size()) exit(7); fclose(f); }
int main(int argc, char **argv) {
  if (argc < 2) return 2;
  if (!strcmp(argv[1], "iqgen") && argc >= 6) {
    const float bal = (float)atof(argv[2]); const int nb = atoi(argv[4]);
    if (bal != 0.0f) gen.setGainBalance(bal);
    std::vector<int16_t> x = slurp(argv[3], (size_t)nb * 128), out((size_t)nb * 256);
    for (int b = 0; b < nb; b++) {
      audio_block_t *bi = new audio_block_t(); memcpy(bi->data, &x[(size_t)b * 128], 256);
      gen.sent[0] = gen.sent[1] = NULL; gen.feed(0, bi); gen.update();
      if (!gen.sent[0] || !gen.sent[1]) { fprintf(stderr, "block %d: nothing transmitted\n", b); return 6; }
      memcpy(&out[(size_t)b * 256], gen.sent[0]->data, 256); memcpy(&out[(size_t)b * 256 + 128], gen.sent[1]->data, 256);
    }
    dump(argv[5], out); return 0;
  }
  if (!strcmp(argv[1], "grab") && argc >= 6) {
    const int nb = atoi(argv[3]), after = atoi(argv[4]);
    std::vector<int16_t> iq = slurp(argv[2], (size_t)nb * 256), out(512, 0);
    for (int b = 0; b < nb; b++) {
      audio_block_t bi, bq; memset(&bi, 0, sizeof bi); memset(&bq, 0, sizeof bq);
      memcpy(bi.data, &iq[(size_t)b * 256], 256); memcpy(bq.data, &iq[(size_t)b * 256 + 128], 256);
      grabber.feed(0, &bi); grabber.feed(1, &bq); grabber.update();
      if (b == after) { printf("new %d\n", (int)grabber.newDataAvailable()); grabber.grab(out.data()); printf("new_after_grab %d\n", (int)grabber.newDataAvailable()); }
    }
    dump(argv[5], out); return 0;
  }
  if (!strcmp(argv[1], "pre") && argc >= 7) {
    const int corr = atoi(argv[2]), swap = atoi(argv[3]), nb = atoi(argv[5]);
    pre.stopAutoI2SerrorDetection();
    pre.setI2SerrorCompensation(corr); pre.swapIQ(swap != 0);
    std::vector<int16_t> iq = slurp(argv[4], (size_t)nb * 256), out((size_t)nb * 256);
    for (int b = 0; b < nb; b++) {
      audio_block_t bi, bq; memset(&bi, 0, sizeof bi); memset(&bq, 0, sizeof bq);
      memcpy(bi.data, &iq[(size_t)b * 256], 256); memcpy(bq.data, &iq[(size_t)b * 256 + 128], 256);
      pre.sent[0] = pre.sent[1] = NULL; pre.feed(0, &bi); pre.feed(1, &bq); pre.update();
      if (!pre.sent[0] || !pre.sent[1]) { fprintf(stderr, "block %d: nothing transmitted\n", b); return 6; }
      memcpy(&out[(size_t)b * 256], pre.sent[0]->data, 256); memcpy(&out[(size_t)b * 256 + 128], pre.sent[1]->data, 256);
    }
    printf("getI2SerrorCompensation %d\ngetAutoI2SerrorDetectionStatus %d\n", (int)pre.getI2SerrorCompensation(), (int)pre.getAutoI2SerrorDetectionStatus());
    dump(argv[6], out); return 0;
  }
  return 2;
}
