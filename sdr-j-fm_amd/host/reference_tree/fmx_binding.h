// fmx_binding.h -- shared by the three replacement sources of the reference tree:
//   src/fm/fm-processor-fmx.cpp      replaces src/fm/fm-processor.cpp
//   src/fm/fm-demodulator-fmx.cpp    replaces src/fm/fm-demodulator.cpp
//   src/rds/rds-decoder-fmx.cpp      replaces src/rds/rds-decoder.cpp
// Every header of the reference stays as it is (includes/fm/fm-processor.h, fm-demodulator.h, includes/rds/rds-decoder.h): radio.cpp
// and the GUI compile unchanged.  The GUI selects the discriminator on ITS fm_Demodulator object (radio.cpp:1687), whose header has no
// getter for the choice, so the replacement setDecoder publishes it here and fmProcessor::run reads it at every block boundary.
#pragma once
#include <atomic>

class fm_Demodulator;

namespace fmx_binding {

// FMX_P_FM_DECODER code (1 AM .. 6 Diff) last selected on demodulator `d`; 3 (Mixed, the reference's default) until set
int  decoder_of(const fm_Demodulator *d);
void publish_decoder(const fm_Demodulator *d, int code);

}  // namespace fmx_binding
