// fmx_rdsgroups.h -- host side of the RDS path behind the bit slicer: block synchronisation and group decoding.
// Pure integer work at 1187.5 bit/s per channel: it stays on the host.  Behaviour follows
//   rdsDecoder::processBit          src/rds/rds-decoder.cpp:104-131
//   rdsBlockSynchronizer            src/rds/rds-blocksynchronizer.cpp:57-336 (constants includes/rds/rds-blocksynchronizer.h:77-91)
//   RDSGroup                        src/rds/rds-group.cpp:33-81
//   rdsGroupDecoder::decode & co.   src/rds/rds-groupdecoder.cpp:71-290
// The Qt signals of those classes (setPiCode, setStationLabel, setRadioText, ...) become fields of fmx_rds_info, which
// the adapter turns back into signals.  The radio text goes through the reference's own character handling (prepareText with its
// alphabet-switch pairs, mapEBUtoUnicode, rds-groupdecoder.cpp:298-343, ebu-codetables.c:65-72; tables in fmx_rds_tables.h) into
// fmx_rds_info::radio_text_ucs2; the raw characters are handed over next to it.
#pragma once
#include <cstdint>
#include <cstring>
#include "../../include/fmx.h"
#include "fmx_rds_tables.h"

namespace fmx {

// mapEBUtoUnicode (ebu-codetables.c:65-72): the alphabet is ignored there, control codes read as a blank
inline uint16_t rds_map_char(uint8_t /*alfabet*/, uint8_t c) { return c < 0x20 ? (uint16_t)' ' : RDS_CHAR_UCS2[c - 0x20]; }
inline const char *rds_pty_name(int pty, int locale) { return (pty < 0 || pty > 31 || locale < 0 || locale > 1) ? nullptr : RDS_PTY_NAME[locale][pty]; }
// rdsGroupDecoder::prepareText (:298-315): the buffer is walked as (previous, current) pairs.  A pair that switches the alphabet
// (alfabetSwitcher :317-325: 0x0F 0x0F, 0x0E 0x0E, 0x1B 0x6E) sets theAlfabet from its first byte (setAlfabetTo :331-343), makes its
// SECOND byte the previous character and steps over the byte behind the pair; otherwise the previous character is emitted.  So the
// last character of `length` never appears, a switch pair leaves its second byte in the text and swallows the character behind it.
// Result trimmed as QString::trimmed does (the table holds no other white space than U+0020).  Returns the number of code units.
inline int rds_prepare_text(const uint8_t *v, int length, uint8_t *alfabet, uint16_t *out, int cap) {
    uint8_t alf = alfabet ? *alfabet : 0;
    uint16_t tmp[256];
    int n = 0;
    if (length > 256) length = 256;
    uint8_t prev = v[0];                                               // (read even when length is 0, as the reference does)
    for (int i = 1; i < length; i++) {
        const uint8_t cur = v[i];
        const bool sw = (prev == 0x0F && cur == 0x0F) || (prev == 0x0E && cur == 0x0E) || (prev == 0x1B && cur == 0x6E);
        if (sw) { alf = prev == 0x0E ? 1 : (prev == 0x1B ? 2 : 0); prev = v[i]; i++; }
        else { tmp[n++] = rds_map_char(alf, prev); prev = cur; }
    }
    if (alfabet) *alfabet = alf;
    int a = 0, e = n;
    while (a < e && tmp[a] == ' ') a++;
    while (e > a && tmp[e - 1] == ' ') e--;
    int k = 0;
    for (int i = a; i < e && k < cap; i++) out[k++] = tmp[i];
    if (k < cap) out[k] = 0;
    return e - a;
}

class RdsGroupDecoderHost {
public:
    RdsGroupDecoderHost() { reset_all(); }
    void reset_all() { sync_reset(); groups_reset(); groups_ok_ = 0; last_type_ = -1; info_ = fmx_rds_info{}; fill_info(); }
    // fmProcessor::resetRds -> rdsDecoder::reset -> rdsGroupDecoder::reset (fm-processor.cpp:862-864, rds-decoder.cpp:65-67,
    // rds-groupdecoder.cpp:71-98): PI, PTY, station label, radio text, M/S and AF go back to "unknown"; the block
    // synchroniser keeps running
    void reset_groups() { groups_reset(); last_type_ = -1; fill_info(); }
    // one sliced bit (rds-decoder.cpp:104-131)
    void push_bit(bool b) {
        switch (push(b)) {
        case WAITING_A: case BUFFERING: break;
        case NO_SYNC: info_.sync_errors = n_sync_err_; resync(); break;
        case NO_CRC: info_.crc_errors = n_crc_err_; resync(); break;
        case COMPLETE: decode_group(); blk_[0] = blk_[1] = blk_[2] = blk_[3] = 0; break;
        }
    }
    const fmx_rds_info &info() { fill_info(); return info_; }

private:
    enum Res { WAITING_A, BUFFERING, NO_SYNC, NO_CRC, COMPLETE };
    static constexpr uint32_t NCRC = 10, NPAY = 16, NBLK = 26, POLY = 0x5B9, REM = 0x31B, BER_RESET = 4000;
    static uint32_t offset_word(int blk, bool typeB) {                 // rds-blocksynchronizer.cpp:197-213
        switch (blk) { default: case 0: return 0xFC; case 1: return 0x198; case 2: return typeB ? 0x350 : 0x168; case 3: return 0x1B4; }
    }
    static uint32_t syndrome(uint32_t bits, uint32_t off) {            // :126-142
        const uint32_t block = bits ^ off;
        uint32_t reg = 0;
        for (int k = (int)NBLK - 1; k >= 0; k--) {
            const uint32_t msb = reg & (1u << (NCRC - 1));
            reg <<= 1;
            if (msb) reg ^= POLY;
            if ((block >> k) & 1u) reg ^= REM;
        }
        return reg;
    }
    bool typeB() const { return ((blk_[1] >> 11) & 1) != 0; }
    void sync_reset() {                                                // :57-70
        stream_ = 0; synced_ = false; cur_ = 0; ber_ = 0.f; bits_in_blk_ = 0; bits_done_ = 0; bit_err_ = 0;
        n_crc_err_ = 0; n_sync_err_ = 0;
        blk_[0] = blk_[1] = blk_[2] = blk_[3] = 0;
    }
    void resync() { cur_ = 0; synced_ = false; bits_in_blk_ = 0; }     // :101-106
    uint32_t meggitt(uint32_t syn) {                                   // :176-195: single-burst correction of the payload
        uint32_t mask = 1u << (NBLK - 1);
        for (uint32_t i = 0; i < NPAY; i++) {
            if (syn & 0x200) {
                if ((syn & 0x1f) == 0) { stream_ ^= mask; bit_err_++; }
                else syn ^= POLY;
            }
            syn <<= 1; mask >>= 1;
        }
        return syn & 0x3FF;
    }
    bool decode_block(int b, uint32_t bits) {                          // :144-173
        uint32_t syn = syndrome(bits, offset_word(b, typeB()));
        if (!synced_) return syn == 0;
        bits_done_ += NPAY;
        if (syn != 0) (void)meggitt(syn);        // (the reference discards doMeggit's result: the block still counts as failed)
        if (syn != 0) bit_err_ += NPAY;
        ber_ = (float)bit_err_ / (float)bits_done_;
        if (bits_done_ >= BER_RESET) { bit_err_ = 0; bits_done_ = 0; }
        return syn == 0;
    }
    Res push(bool b) {                                                 // :215-336
        stream_ = (stream_ << 1) | (b ? 1u : 0u);
        if (synced_) {
            if (++bits_in_blk_ < NBLK) return BUFFERING;
            bits_in_blk_ = 0;
            if (!decode_block(cur_, stream_)) { n_crc_err_++; return NO_CRC; }
            blk_[cur_] = (uint16_t)(stream_ >> NCRC);
            const Res r = cur_ == 3 ? COMPLETE : BUFFERING;
            cur_ = (cur_ + 1) & 3;
            return r;
        }
        if (cur_ == 0) {                                               // slide bit by bit until a clean block A appears
            if (syndrome(stream_ & 0x3FFFFFF, offset_word(0, typeB())) != 0) return WAITING_A;
            blk_[0] = (uint16_t)(stream_ >> NCRC);
            bits_in_blk_ = 0; cur_ = 1;
            return BUFFERING;
        }
        if (bits_in_blk_ < NBLK - 1) { bits_in_blk_++; return BUFFERING; }
        bits_in_blk_ = 0;
        if (syndrome(stream_, offset_word(cur_, typeB())) != 0) { n_sync_err_++; return NO_SYNC; }
        blk_[cur_] = (uint16_t)(stream_ >> NCRC);
        if (cur_ < 2) { cur_++; return BUFFERING; }                    // SYNC_END_BLOCK = BLOCK_C
        synced_ = true;
        const Res r = cur_ == 3 ? COMPLETE : BUFFERING;
        cur_ = (cur_ + 1) & 3;
        return r;
    }
    // ---- group decoder (rds-groupdecoder.cpp)
    void groups_reset() {                                              // :71-98
        pi_ = 0; pty_ = -1;
        std::memset(ps_, ' ', 8); ps_[8] = 0; ps_seg_ = 0; di_ = 0;
        std::memset(rt_, ' ', 64); rt_[64] = 0; rt_ab_ = -1; rt_seg_ = 0; rt_len_ = 0; rt_shown_[0] = 0;
        ms_ = -1; af1_ = af2_ = 0;
        rt_u16_[0] = 0; rt_u16_len_ = 0;                               // clearRadioText (:95); theAlfabet is not touched by reset ()
    }
    void show_text(int len) {                                          // prepareText :298-315; rt_shown_ = the same characters unmapped
        // the reference emits characters v[0 .. len-2] (it walks pairs, dropping the last one) and trims the result
        int n = len - 1; if (n < 0) n = 0;
        auto sp = [](char c) { return c == ' ' || (c >= 0x09 && c <= 0x0D); };     // QString::trimmed
        int a = 0; while (a < n && sp(rt_[a])) a++;
        int e = n; while (e > a && sp(rt_[e - 1])) e--;
        std::memcpy(rt_shown_, rt_ + a, (size_t)(e - a)); rt_shown_[e - a] = 0; rt_len_ = e - a;
        rt_u16_len_ = rds_prepare_text(reinterpret_cast<const uint8_t *>(rt_), len, &alfabet_, rt_u16_, 65);
    }
    void decode_group() {                                              // decode :100-165
        groups_ok_++;
        last_type_ = (blk_[1] >> 12) & 0xF;
        if (blk_[0] != pi_) { groups_reset(); pi_ = blk_[0]; }
        pty_ = (blk_[1] >> 5) & 0x1F;
        if (typeB()) return;                                           // "Cannot decode B type groups"
        if (last_type_ == 0) {                                         // Handle_Basic_Tuning_and_Switching :167-180
            const uint32_t seg = blk_[1] & 3;
            ps_[2 * seg] = (char)(blk_[3] >> 8); ps_[2 * seg + 1] = (char)(blk_[3] & 0xFF);
            if (seg == 0) ps_seg_ = 0;
            ps_seg_ |= 2 * seg;
            const uint8_t a1 = (uint8_t)(blk_[2] >> 8), a2 = (uint8_t)(blk_[2] & 0xFF);            // additionalFrequencies :207-218
            af1_ = (a1 > 0 && a1 < 205) ? a1 * 100 + 87500 : 0;
            af2_ = (a1 != 250 && a2 > 0 && a2 < 205) ? a2 * 100 + 87500 : 0;
            ms_ = (blk_[1] >> 3) & 1;
            di_ |= ((blk_[1] >> 2) & 1) << seg;
        } else if (last_type_ == 2) {                                  // Handle_RadioText :222-281
            const int ab = (blk_[1] >> 4) & 1; const int seg = blk_[1] & 0xF;
            if (rt_ab_ != ab) { rt_ab_ = ab; rt_seg_ = 0; std::memset(rt_, ' ', 64); rt_[64] = 0; rt_shown_[0] = 0; rt_len_ = 0; rt_u16_[0] = 0; rt_u16_len_ = 0; }
            char *f = &rt_[4 * seg];
            f[0] = (char)(blk_[2] >> 8); f[1] = (char)(blk_[2] & 0xFF); f[2] = (char)(blk_[3] >> 8); f[3] = (char)(blk_[3] & 0xFF);
            rt_seg_ |= 1u << seg;
            if (rt_seg_ + 1 == (1u << (seg + 1))) show_text(seg * 4);
            bool end = false;
            for (int i = 0; i < 4; i++) if (f[i] == 0x0D) end = true;
            if (end || rt_seg_ + 1 == (1u << 16)) show_text(64);
        }
    }
    void fill_info() {
        info_.synchronized = synced_ ? 1 : 0; info_.pi_code = pi_; info_.pty_code = pty_; info_.last_group_type = last_type_;
        info_.groups_decoded = groups_ok_; info_.crc_errors = n_crc_err_; info_.sync_errors = n_sync_err_; info_.bit_error_rate = ber_;
        std::memcpy(info_.station_label, ps_, 9); std::memcpy(info_.radio_text, rt_shown_, 65);
        info_.af1_khz = af1_; info_.af2_khz = af2_; info_.music_speech = ms_; info_.di_code = di_;
        std::memcpy(info_.radio_text_ucs2, rt_u16_, sizeof(rt_u16_)); info_.radio_text_ucs2_len = (int16_t)rt_u16_len_;
    }
    // synchroniser
    uint32_t stream_; bool synced_; int cur_; float ber_; uint32_t bits_in_blk_, bits_done_, bit_err_; int n_crc_err_, n_sync_err_;
    uint16_t blk_[4];
    // groups
    int32_t pi_, pty_, last_type_ = -1, groups_ok_ = 0, ms_, af1_, af2_; uint32_t ps_seg_, di_, rt_seg_; int rt_ab_, rt_len_;
    char ps_[9], rt_[65], rt_shown_[65];
    uint16_t rt_u16_[65] = {0}; int rt_u16_len_ = 0; uint8_t alfabet_ = 0;
    fmx_rds_info info_;
};

}  // namespace fmx
