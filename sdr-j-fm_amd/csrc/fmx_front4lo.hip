// fmx_front4lo.hip -- stage A on the matrix pipe for handles with LOCAL OSCILLATORS: fmx_front4.hip compiled a second time, twelve waves on one channel per
// workgroup and complex taps (see the note at the top of that file).  BASELINE configs[2]: 256 carriers in 24 shared wide-band streams.
#define F4_NS f4lo
#define F4_NW 12
#define F4_CPW 1
#define F4_LO 1
#define F4_FN(name) name##_lo
#include "fmx_front4.hip"
