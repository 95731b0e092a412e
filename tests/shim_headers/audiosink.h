// compile-check stand-in for includes/output/audiosink.h:36-76 (the real one needs portaudio.h): the calls fmProcessor makes
#pragma once
#include "fm-constants.h"
class audioSink {
public:
    int32_t putSample(DSPCOMPLEX);
    int32_t putSamples(DSPCOMPLEX *, int32_t);
};
