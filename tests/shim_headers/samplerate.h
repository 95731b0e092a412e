// compile-check stand-in: the libsamplerate names includes/various/newconverter.h mentions
#pragma once
extern "C" {
typedef struct SRC_STATE_tag SRC_STATE;
typedef struct { const float *data_in; float *data_out; long input_frames, output_frames, input_frames_used, output_frames_gen;
                 int end_of_input; double src_ratio; } SRC_DATA;
enum { SRC_SINC_BEST_QUALITY = 0, SRC_SINC_MEDIUM_QUALITY = 1, SRC_SINC_FASTEST = 2, SRC_ZERO_ORDER_HOLD = 3, SRC_LINEAR = 4 };
SRC_STATE *src_new(int, int, int *);
SRC_STATE *src_delete(SRC_STATE *);
int src_process(SRC_STATE *, SRC_DATA *);
const char *src_strerror(int);
int src_reset(SRC_STATE *);
}
