// TEST-ONLY stand-in for the *host application's* AudioStream.h, used to compile and exercise
// include/AudioSDR_hip.hpp and include/AudioSDRlib_hip.hpp (this project's adaptors).  It is NOT used to build any part of the reference.
#ifndef TEST_MOCK_AUDIOSTREAM_H_
#define TEST_MOCK_AUDIOSTREAM_H_
#include <stdint.h>
#define AUDIO_BLOCK_SAMPLES 128
struct audio_block_t { uint8_t ref_count, reserved1; uint16_t memory_pool_index; int16_t data[AUDIO_BLOCK_SAMPLES]; };
class AudioStream {
 public:
  AudioStream(unsigned char ninput, audio_block_t **iqueue) : n_(ninput), q_(iqueue) { for (int i = 0; i < ninput; i++) q_[i] = nullptr; out_[0] = out_[1] = nullptr; released_ = 0; }
  virtual ~AudioStream() {}
  virtual void update(void) = 0;
  void feed(unsigned ch, audio_block_t *b) { q_[ch] = b; }      // test hook
  audio_block_t *out_[2]; int released_;
  audio_block_t *spare_ = nullptr;                              // test hook: what allocate() hands out
 protected:
  audio_block_t *receiveWritable(unsigned ch) { audio_block_t *b = q_[ch]; q_[ch] = nullptr; return b; }
  audio_block_t *receiveReadOnly(unsigned ch) { return receiveWritable(ch); }
  audio_block_t *allocate(void) { return spare_; }
  void transmit(audio_block_t *b, unsigned char ch) { out_[ch] = b; }
  void release(audio_block_t *) { released_++; }
 private:
  unsigned char n_; audio_block_t **q_;
};
#endif
