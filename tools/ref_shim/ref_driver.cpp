// Driver of the stand-in build (tools/ref_shim/README.md): ONE AudioSDR instance of the reference's own source (static storage; one instance per
// process: the reference's function-statics), configured by a script of setter calls, fed blocks of I/Q, its output blocks and getters
// written out.   ref_driver <script.txt> <iq.bin int16 [blocks][2][128]> <n_blocks> <out.bin int16 [blocks][128]>  -> getters on stdout
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <string>
#include "AudioSDR.h"
SerialShim Serial;
static AudioSDR sdr;
static int call(const char *m, const double *a, int na) {
#define M0(name) if (!strcmp(m, #name)) { sdr.name(); return 0; }
#define M1(name, T) if (!strcmp(m, #name)) { if (na < 1) return -1; sdr.name((T)a[0]); return 0; }
  M0(init) M1(setMute, bool) M1(setInputGain, float) M1(setIQgainBalance, float) M1(setDemodMode, int)
  M0(enableAudioFilter) M0(disableAudioFilter) M1(setOutputGain, float) M1(setAudioFilter, int)
  M0(enableALSfilter) M0(disableALSfilter) M0(setALSfilterNotch) M0(setALSfilterPeak) M0(setALSfilterAdaptive) M0(setALSfilterStatic)
  if (!strcmp(m, "setALSfilterParams")) { if (na < 3) return -1; sdr.setALSfilterParams((unsigned int)a[0], (float)a[1], (float)a[2]); return 0; }
  M0(enableAGC) M0(disableAGC) M1(setAGCthreshold, float) M1(setAGCslope, float) M1(setAGCmode, int16_t) M1(setAGCkneeWidth, float)
  M1(setAGCattackTime, float) M1(setAGCreleaseTime, float) M1(setAGChangTime, float) M1(setAGCstaticGain, float)
  M0(enableNoiseBlanker) M0(disableNoiseBlanker) M1(setNoiseBlankerThreshold, float) M1(setNoiseBlankerThresholdDb, float)
  return -2;
}
int main(int argc, char **argv) {
  if (argc < 5) return 2;
  const int nb = atoi(argv[3]);
  std::vector<int16_t> iq((size_t)nb * 256), out((size_t)nb * 128);
  FILE *fi = fopen(argv[2], "rb");
  if (!fi || fread(iq.data(), 2, iq.size(), fi) != iq.size()) return 5;
  fclose(fi);
  FILE *fs = fopen(argv[1], "r");
  if (!fs) return 3;
  char line[512];
  int b = 0;   // next block to feed
  auto feed_blocks = [&](int upto) -> int {
  for (; b < upto && b < nb; b++) {
    audio_block_t bi, bq;
    memset(&bi, 0, sizeof bi); memset(&bq, 0, sizeof bq);
    memcpy(bi.data, &iq[(size_t)b * 256], 256); memcpy(bq.data, &iq[(size_t)b * 256 + 128], 256);
    sdr.sent[0] = sdr.sent[1] = NULL;
    sdr.feed(0, &bi); sdr.feed(1, &bq);
    sdr.update();
    if (!sdr.sent[0] || sdr.sent[0] != sdr.sent[1]) { fprintf(stderr, "block %d: update() transmitted nothing / different blocks\n", b); return 6; }
    memcpy(&out[(size_t)b * 128], sdr.sent[0]->data, 256);
  }
  return 0; };
  // the script: setter calls, and "run <k>" = feed the next k blocks (setters between blocks); whatever is left is fed at the end
  while (fgets(line, sizeof line, fs)) {
    char m[128]; double a[4] = {0, 0, 0, 0};
    const int n = sscanf(line, "%127s %lf %lf %lf %lf", m, &a[0], &a[1], &a[2], &a[3]);
    if (n < 1) continue;
    if (!strcmp(m, "run")) { const int rc = feed_blocks(b + (int)a[0]); if (rc) return rc; continue; }
    if (call(m, a, n - 1) != 0) { fprintf(stderr, "bad script line: %s", line); return 4; }
  }
  fclose(fs);
  { const int rc = feed_blocks(nb); if (rc) return rc; }
  FILE *fo = fopen(argv[4], "wb");
  if (!fo || fwrite(out.data(), 2, out.size(), fo) != out.size()) return 7;
  fclose(fo);
  auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
  printf("getDemodMode %d\n", (int)sdr.getDemodMode());
  printf("getTuningOffset %08x\ngetBPFlower %08x\ngetBPFupper %08x\n", bits(sdr.getTuningOffset()), bits(sdr.getBPFlower()), bits(sdr.getBPFupper()));
  printf("getMute %d\ngetAudioFilter %d\n", (int)sdr.getMute(), (int)sdr.getAudioFilter());
  printf("ALSfilterIsEnabled %d\nALSfilterIsNotch %d\nALSfilterIsPeak %d\nALSfilterIsAdaptive %d\n", (int)sdr.ALSfilterIsEnabled(), (int)sdr.ALSfilterIsNotch(), (int)sdr.ALSfilterIsPeak(), (int)sdr.ALSfilterIsAdaptive());
  printf("AGCisEnabled %d\nAGCisActive %d\n", (int)sdr.AGCisEnabled(), (int)sdr.AGCisActive());
  printf("getAGCthreshold %08x\ngetAGCslope %08x\ngetAGCkneeWidth %08x\ngetAGCattack %08x\ngetAGCrelease %08x\n", bits(sdr.getAGCthreshold()), bits(sdr.getAGCslope()),
         bits(sdr.getAGCkneeWidth()), bits(sdr.getAGCattack()), bits(sdr.getAGCrelease()));
  printf("getAAGalphaAttack %08x\ngetAGCbetaAttack %08x\ngetAGCalphaRelease %08x\ngetAGCbetaRelease %08x\ngetAGCstaticGain %08x\n", bits(sdr.getAAGalphaAttack()),
         bits(sdr.getAGCbetaAttack()), bits(sdr.getAGCalphaRelease()), bits(sdr.getAGCbetaRelease()), bits(sdr.getAGCstaticGain()));
  printf("getAMcarrierLevel %08x\n", bits(sdr.getAMcarrierLevel()));
  printf("NoiseBlankerisEnabled %d\nNoiseBlankerDetection %d\n", (int)sdr.NoiseBlankerisEnabled(), (int)sdr.NoiseBlankerDetection());
  printf("getSAMfrequency %08x\ngetSAMphaseLockStatus %d\n", bits(sdr.getSAMfrequency()), (int)sdr.getSAMphaseLockStatus());
  printf("getAGClookup");
  for (int i = 0; i <= 128; i++) printf(" %08x", bits(sdr.getAGClookup(i)));
  printf("\n");
  return 0;
}
