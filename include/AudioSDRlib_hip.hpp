// AudioSDRlib_hip.hpp -- header-only drop-ins for the other AudioStream classes of the reference library
// (SRC/AudioSDRlib/AudioSDRlib.h bundles them with AudioSDR), over the C ABI of libasdr_hip.so (asdr_front.h),
// one channel per object:
//   class AudioSDRpreProcessor    (AudioSDRpreProcessor.h:49-84,    update(): AudioSDRpreProcessor.cpp:46-138)
//   class AudioIQgenerator        (AudioIQgenerator.h:48-106,       update(): AudioIQgenerator.cpp:33-87)
//   class AudioGrabberComplex256  (AudioGrabberComplex256.h:44-63,  update(): AudioGrabberComplex256.cpp:50-72)
// Same class names, public methods and block traffic (receiveWritable / receiveReadOnly / allocate / transmit /
// release) as the reference.  The application provides "AudioStream.h" as for AudioSDR_hip.hpp; in addition to the
// members listed there these classes use receiveReadOnly(unsigned) and allocate(), as the reference does.
// One channel per object wastes the GPU: N receivers should share the batched objects.
#ifndef AUDIOSDRLIB_HIP_HPP_
#define AUDIOSDRLIB_HIP_HPP_

#include "AudioSDR_hip.hpp"
#include "asdr_front.h"

class AudioSDRpreProcessor : public AudioStream {
 public:
  explicit AudioSDRpreProcessor(int hip_device = 0) : AudioStream(2, inputQueueArray), p_(asdr_pre_create(1, hip_device)) {}
  ~AudioSDRpreProcessor() { asdr_pre_destroy(p_); }
  AudioSDRpreProcessor(const AudioSDRpreProcessor &) = delete;
  AudioSDRpreProcessor &operator=(const AudioSDRpreProcessor &) = delete;
  bool ok() const { return p_ != nullptr; }

  virtual void update(void) {                                       // AudioSDRpreProcessor.cpp:46-138
    audio_block_t *blockI = receiveWritable(0), *blockQ = receiveWritable(1);
    if (!blockI && blockQ) { release(blockQ); return; }             // :50-52
    if (blockI && !blockQ) { release(blockI); return; }
    if (!blockI && !blockQ) return;
    asdr_pre_update(p_, blockI->data, blockQ->data, 1);             // in place, like the reference
    transmit(blockI, 0);                                            // :134-137
    transmit(blockQ, 1);
    release(blockQ);
    release(blockI);
  }
  void startAutoI2SerrorDetection(void) { asdr_pre_startAutoI2SerrorDetection(p_, 0); }
  void stopAutoI2SerrorDetection(void) { asdr_pre_stopAutoI2SerrorDetection(p_, 0); }
  bool getAutoI2SerrorDetectionStatus(void) { return asdr_pre_getAutoI2SerrorDetectionStatus(p_, 0) != 0; }
  void setI2SerrorCompensation(int correction) { asdr_pre_setI2SerrorCompensation(p_, 0, correction); }
  int16_t getI2SerrorCompensation(void) { return asdr_pre_getI2SerrorCompensation(p_, 0); }
  void swapIQ(bool swap) { asdr_pre_swapIQ(p_, 0, swap); }

 private:
  audio_block_t *inputQueueArray[2];
  asdr_pre_t *p_;
};

class AudioIQgenerator : public AudioStream {
 public:
  explicit AudioIQgenerator(int hip_device = 0) : AudioStream(1, inputQueueArray), g_(asdr_iqgen_create(1, hip_device)) {}
  ~AudioIQgenerator() { asdr_iqgen_destroy(g_); }
  AudioIQgenerator(const AudioIQgenerator &) = delete;
  AudioIQgenerator &operator=(const AudioIQgenerator &) = delete;
  bool ok() const { return g_ != nullptr; }

  virtual void update(void) {                                       // AudioIQgenerator.cpp:33-87
    audio_block_t *blockI = receiveWritable(0);
    if (!blockI) return;                                            // :43-45
    audio_block_t *blockQ = allocate();
    if (!blockQ) return;                                            // :47-49 (the reference keeps blockI here too)
    asdr_iqgen_update(g_, blockI->data, blockI->data, blockQ->data, 1);
    transmit(blockI, 0);                                            // :83-86
    release(blockI);
    transmit(blockQ, 1);
    release(blockQ);
  }
  void setGainBalance(float balance) { asdr_iqgen_setGainBalance(g_, 0, balance); }

 private:
  audio_block_t *inputQueueArray[1];
  asdr_iqgen_t *g_;
};

class AudioGrabberComplex256 : public AudioStream {
 public:
  explicit AudioGrabberComplex256(int hip_device = 0) : AudioStream(2, inputQueueArray), g_(asdr_grab_create(1, hip_device)) {}
  ~AudioGrabberComplex256() { asdr_grab_destroy(g_); }
  AudioGrabberComplex256(const AudioGrabberComplex256 &) = delete;
  AudioGrabberComplex256 &operator=(const AudioGrabberComplex256 &) = delete;
  bool ok() const { return g_ != nullptr; }

  virtual void update(void) {                                       // AudioGrabberComplex256.cpp:50-72
    audio_block_t *blockI = receiveReadOnly(0), *blockQ = receiveReadOnly(1);
    if (!blockI && blockQ) { release(blockQ); return; }
    if (blockI && !blockQ) { release(blockI); return; }
    if (!blockI && !blockQ) return;
    asdr_grab_update(g_, blockI->data, blockQ->data, 1);
    release(blockI);
    release(blockQ);
  }
  bool newDataAvailable(void) { return asdr_grab_newDataAvailable(g_, 0) != 0; }
  void grab(int16_t *destination) { asdr_grab_grab(g_, 0, destination); }

 private:
  audio_block_t *inputQueueArray[2];
  asdr_grab_t *g_;
};

#endif  // AUDIOSDRLIB_HIP_HPP_
