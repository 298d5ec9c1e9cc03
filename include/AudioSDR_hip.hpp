// AudioSDR_hip.hpp -- header-only drop-in for the reference's `class AudioSDR : public AudioStream`
// (SRC/AudioSDRlib/AudioSDR.h:75-156) on a host with an MI355X: same class name, same 59 public methods, same
// update() contract, implemented over the C ABI of libasdr_hip.so (asdr.h) with a one-channel batch.
//
// The application provides its own "AudioStream.h" (the PJRC Teensy audio-library types or its PC port):
//   struct audio_block_t { ...; int16_t data[AUDIO_BLOCK_SAMPLES]; };   AUDIO_BLOCK_SAMPLES == 128
//   class AudioStream { protected: audio_block_t* receiveWritable(unsigned); void transmit(audio_block_t*, unsigned char);
//                       static void release(audio_block_t*); public: AudioStream(unsigned char, audio_block_t**); virtual void update() = 0; };
// Only the members the reference itself uses are needed (AudioSDR.cpp:46-56, 158-167).
//
// One channel per object wastes the GPU; N receivers should share one asdr_batch_t.
#ifndef AUDIOSDR_HIP_HPP_
#define AUDIOSDR_HIP_HPP_

#include "AudioStream.h"
#include "asdr.h"

#ifndef float32_t
typedef float float32_t;
#endif

// enumerators keep the reference's values (AudioSDR.h:44-71)
#define LSBmode 0
#define USBmode 1
#define CW_LSBmode 2
#define CW_USBmode 3
#define AMmode 4
#define SAMmode 5
#define WSPRmode 6
#define audioAM 0
#define audioCW 1
#define audioWSPR 2
#define audio2100 3
#define audio2300 4
#define audio2500 5
#define audio2700 6
#define audio2900 7
#define audio3100 8
#define audio3300 9
#define audioBypass 10
#define AGCoff 0
#define AGCfast 1
#define AGCmedium 2
#define AGCslow 3

class AudioSDR : public AudioStream {
 public:
  explicit AudioSDR(int hip_device = 0) : AudioStream(2, inputQueueArray), b_(asdr_create(1, hip_device)) {}
  ~AudioSDR() { asdr_destroy(b_); }
  AudioSDR(const AudioSDR &) = delete;
  AudioSDR &operator=(const AudioSDR &) = delete;
  bool ok() const { return b_ != nullptr; }

  virtual void update(void) {                            // AudioSDR.cpp:39-168
    audio_block_t *blockI = receiveWritable(0), *blockQ = receiveWritable(1);
    if (!blockI && blockQ) { release(blockQ); return; }  // missing-input guard, :48-56
    if (blockI && !blockQ) { release(blockI); return; }
    if (!blockI && !blockQ) return;
    asdr_update(b_, blockI->data, blockQ->data, blockI->data, 1);   // mono audio reuses blockI, :158-161
    transmit(blockI, 0);                                 // :164-165
    transmit(blockI, 1);
    release(blockI);                                     // :166-167
    release(blockQ);
  }
  // --- general (AudioSDR.h:88-97)
  void init(void) { asdr_init(b_, 0); }
  void setMute(bool m) { asdr_setMute(b_, 0, m); }
  void setInputGain(float g) { asdr_setInputGain(b_, 0, g); }
  void setIQgainBalance(float v) { asdr_setIQgainBalance(b_, 0, v); }
  int16_t getDemodMode(void) { return asdr_getDemodMode(b_, 0); }
  float32_t setDemodMode(int m) { return asdr_setDemodMode(b_, 0, m); }
  float32_t getBPFlower(void) { return asdr_getBPFlower(b_, 0); }
  float32_t getBPFupper(void) { return asdr_getBPFupper(b_, 0); }
  float32_t getTuningOffset(void) { return asdr_getTuningOffset(b_, 0); }
  bool getMute(void) { return asdr_getMute(b_, 0) != 0; }
  // --- IIR audio output filters (AudioSDR.h:100-104)
  void enableAudioFilter(void) { asdr_enableAudioFilter(b_, 0); }
  void disableAudioFilter(void) { asdr_disableAudioFilter(b_, 0); }
  int getAudioFilter(void) { return asdr_getAudioFilter(b_, 0); }
  void setOutputGain(float g) { asdr_setOutputGain(b_, 0, g); }
  void setAudioFilter(int f) { asdr_setAudioFilter(b_, 0, f); }
  // --- ALS notch/peaking filter (AudioSDR.h:107-117)
  void enableALSfilter(void) { asdr_enableALSfilter(b_, 0); }
  void disableALSfilter(void) { asdr_disableALSfilter(b_, 0); }
  void setALSfilterNotch(void) { asdr_setALSfilterNotch(b_, 0); }
  void setALSfilterPeak(void) { asdr_setALSfilterPeak(b_, 0); }
  void setALSfilterAdaptive(void) { asdr_setALSfilterAdaptive(b_, 0); }
  void setALSfilterStatic(void) { asdr_setALSfilterStatic(b_, 0); }
  void setALSfilterParams(unsigned int m, float lambda, float delay) { asdr_setALSfilterParams(b_, 0, m, lambda, delay); }
  bool ALSfilterIsEnabled(void) { return asdr_ALSfilterIsEnabled(b_, 0) != 0; }
  bool ALSfilterIsNotch(void) { return asdr_ALSfilterIsNotch(b_, 0) != 0; }
  bool ALSfilterIsPeak(void) { return asdr_ALSfilterIsPeak(b_, 0) != 0; }
  bool ALSfilterIsAdaptive(void) { return asdr_ALSfilterIsAdaptive(b_, 0) != 0; }
  // --- AGC processor (AudioSDR.h:120-144; getAGCmakeUpMode is declared but never defined upstream)
  void enableAGC(void) { asdr_enableAGC(b_, 0); }
  void disableAGC(void) { asdr_disableAGC(b_, 0); }
  bool AGCisEnabled(void) { return asdr_AGCisEnabled(b_, 0) != 0; }
  bool AGCisActive(void) { return asdr_AGCisActive(b_, 0) != 0; }
  void setAGCthreshold(float v) { asdr_setAGCthreshold(b_, 0, v); }
  void setAGCslope(float v) { asdr_setAGCslope(b_, 0, v); }
  void setAGCmode(int16_t m) { asdr_setAGCmode(b_, 0, m); }
  void setAGCkneeWidth(float v) { asdr_setAGCkneeWidth(b_, 0, v); }
  void setAGCattackTime(float ms) { asdr_setAGCattackTime(b_, 0, ms); }
  void setAGCreleaseTime(float ms) { asdr_setAGCreleaseTime(b_, 0, ms); }
  void setAGChangTime(float ms) { asdr_setAGChangTime(b_, 0, ms); }
  void setAGCstaticGain(float g) { asdr_setAGCstaticGain(b_, 0, g); }
  float32_t getAGCthreshold(void) { return asdr_getAGCthreshold(b_, 0); }
  float32_t getAGCslope(void) { return asdr_getAGCslope(b_, 0); }
  float32_t getAGCkneeWidth(void) { return asdr_getAGCkneeWidth(b_, 0); }
  float32_t getAGCattack(void) { return asdr_getAGCattack(b_, 0); }
  float32_t getAGCrelease(void) { return asdr_getAGCrelease(b_, 0); }
  float32_t getAAGalphaAttack(void) { return asdr_getAAGalphaAttack(b_, 0); }
  float32_t getAGCbetaAttack(void) { return asdr_getAGCbetaAttack(b_, 0); }
  float32_t getAGCalphaRelease(void) { return asdr_getAGCalphaRelease(b_, 0); }
  float32_t getAGCbetaRelease(void) { return asdr_getAGCbetaRelease(b_, 0); }
  float32_t getAGClookup(int i) { return asdr_getAGClookup(b_, 0, i); }
  float32_t getAGCstaticGain(void) { return asdr_getAGCstaticGain(b_, 0); }
  float32_t getAMcarrierLevel(void) { return asdr_getAMcarrierLevel(b_, 0); }
  // --- impulse noise blanker (AudioSDR.h:147-152)
  void enableNoiseBlanker(void) { asdr_enableNoiseBlanker(b_, 0); }
  void disableNoiseBlanker(void) { asdr_disableNoiseBlanker(b_, 0); }
  void setNoiseBlankerThreshold(float r) { asdr_setNoiseBlankerThreshold(b_, 0, r); }
  void setNoiseBlankerThresholdDb(float db) { asdr_setNoiseBlankerThresholdDb(b_, 0, db); }
  bool NoiseBlankerisEnabled(void) { return asdr_NoiseBlankerisEnabled(b_, 0) != 0; }
  bool NoiseBlankerDetection(void) { return asdr_NoiseBlankerDetection(b_, 0) != 0; }
  // --- synchronous AM detector (AudioSDR.h:155-156)
  float32_t getSAMfrequency(void) { return asdr_getSAMfrequency(b_, 0); }
  bool getSAMphaseLockStatus(void) { return asdr_getSAMphaseLockStatus(b_, 0) != 0; }

 private:
  audio_block_t *inputQueueArray[2];
  asdr_batch_t *b_;
};

#endif  // AUDIOSDR_HIP_HPP_
