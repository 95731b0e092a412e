// src/rds/rds-decoder-fmx.cpp -- replaces src/rds/rds-decoder.cpp in the reference tree when fmProcessor runs on libfmx.
// includes/rds/rds-decoder.h is unchanged, and so are rds-blocksynchronizer.cpp, rds-group.cpp, rds-groupdecoder.cpp and
// ebu-codetables.c: the three bit slicers (rds-decoder-1/2/3.cpp) run on the GPU (csrc/fmx_rds.hip), every bit they decide comes
// back through fmx_rds_bits and goes through the REFERENCE'S OWN block synchroniser and group decoder here -- what the GUI shows
// (PI, PTY name, station label, radio text, AF, M/S, error counters) is produced by the reference's code from the same bits.
#include "rds-decoder.h"
#include "radio.h"

// (the slicer objects are not created: their work is done on the GPU)
rdsDecoder::rdsDecoder(RadioInterface *myRadioInterface, int32_t rate)
    : my_rdsGroupDecoder(myRadioInterface), my_rdsBlockSync(myRadioInterface),
      my_costas(rate, 1.0f / 16.0f, 0.02f / 16.0f, 10.0f), my_AGC(2e-3f, 0.38f, 9.0f) {
    decoder_1 = nullptr; decoder_2 = nullptr; decoder_3 = nullptr;
    my_rdsGroup.clear();
    my_rdsBlockSync.setFecEnabled(true);
    connect(this, SIGNAL(setCRCErrors(int)), myRadioInterface, SLOT(setCRCErrors(int)));
    connect(this, SIGNAL(setSyncErrors(int)), myRadioInterface, SLOT(setSyncErrors(int)));
}
rdsDecoder::~rdsDecoder() {}

void rdsDecoder::reset() { my_rdsGroupDecoder.reset(); }

// fmProcessor::run (fm-processor-fmx.cpp) calls this once per bit the GPU slicer decided: `v` = the constellation point the bit was
// decided on (fmx_rds_symbols; it goes back to the caller in *m for the IQ scope, as rds-decoder-2.cpp:108-114 does).  The header's
// signature has no room for the bit itself and must stay as it is, so the bit rides in bit 8 of the last argument:
// ptyLocale is 0 or 1, so the caller passes (ptyLocale | bit << 8) and the header stays untouched.
bool rdsDecoder::doDecode(const DSPCOMPLEX v, DSPCOMPLEX *const m, ERdsMode mode, int ptyLocaleAndBit) {
    if (mode == ERdsMode::RDS_OFF) return false;
    *m = v;
    processBit(((ptyLocaleAndBit >> 8) & 1) != 0, ptyLocaleAndBit & 0xFF);
    return true;
}

// one bit into the block synchroniser; a complete group into the group decoder (the reference's flow: sync errors and CRC errors are
// reported and followed by a resynchronisation, a complete group is decoded and cleared)
void rdsDecoder::processBit(bool bit, int ptyLocale) {
    const rdsBlockSynchronizer::SyncResult r = my_rdsBlockSync.pushBit(bit, &my_rdsGroup);
    if (r == rdsBlockSynchronizer::RDS_NO_SYNC) {
        setSyncErrors(my_rdsBlockSync.getNumSyncErrors());
        my_rdsBlockSync.resync();
    } else if (r == rdsBlockSynchronizer::RDS_NO_CRC) {
        setCRCErrors(my_rdsBlockSync.getNumCRCErrors());
        my_rdsBlockSync.resync();
    } else if (r == rdsBlockSynchronizer::RDS_COMPLETE_GROUP) {
        (void)my_rdsGroupDecoder.decode(&my_rdsGroup, ptyLocale);
        my_rdsGroup.clear();
    }
}
