// compile-check stand-in: the libsndfile names the reference's headers and the binding mention
#pragma once
#include <stdint.h>
extern "C" {
typedef struct sf_private_tag SNDFILE;
typedef int64_t sf_count_t;
typedef struct { sf_count_t frames; int samplerate, channels, format, sections, seekable; } SF_INFO;
sf_count_t sf_writef_float(SNDFILE *, const float *, sf_count_t);
}
