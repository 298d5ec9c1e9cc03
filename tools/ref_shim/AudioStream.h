// STAND-IN for the Teensy audio library's AudioStream.h (tools/ref_shim/README.md): a queue-backed AudioStream of exactly the members
// AudioSDR uses (AudioSDR.cpp:46-56, 164-167).
#pragma once
#include <stdint.h>
#include <stddef.h>
#define AUDIO_BLOCK_SAMPLES 128
#define AUDIO_SAMPLE_RATE_EXACT 44100.0f
#define AUDIO_SAMPLE_RATE AUDIO_SAMPLE_RATE_EXACT
typedef struct audio_block_struct {
  uint8_t ref_count, reserved1;
  uint16_t memory_pool_index;
  int16_t data[AUDIO_BLOCK_SAMPLES];
} audio_block_t;
class AudioStream {
 public:
  AudioStream(unsigned char ninput, audio_block_t **iqueue) : num_inputs(ninput), inputQueue(iqueue) {
    for (int i = 0; i < ninput; i++) inputQueue[i] = NULL;
    for (int i = 0; i < 4; i++) sent[i] = NULL;
  }
  virtual void update(void) = 0;
  // test side
  void feed(unsigned int ch, audio_block_t *b) { inputQueue[ch] = b; }
  audio_block_t *sent[4];
  int released = 0;
 protected:
  audio_block_t *receiveWritable(unsigned int index = 0) { if (index >= num_inputs) return NULL; audio_block_t *b = inputQueue[index]; inputQueue[index] = NULL; return b; }
  audio_block_t *receiveReadOnly(unsigned int index = 0) { return receiveWritable(index); }
  void transmit(audio_block_t *block, unsigned char index = 0) { if (index < 4) sent[index] = block; }
  void release(audio_block_t *) { released++; }
  static audio_block_t *allocate(void) { return new audio_block_t(); }
  unsigned char num_inputs;
  audio_block_t **inputQueue;
};
