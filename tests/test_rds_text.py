"""RDS byte work pinned to the reference: RDSGroup's field extraction (src/rds/rds-group.cpp:33-81), the
programme-type names and the character map of src/rds/ebu-codetables.c, and rdsGroupDecoder::prepareText with its alphabet-switch pairs
(src/rds/rds-groupdecoder.cpp:298-343).

Three legs, all CPU:
  * tests/golden/ref_rds_tables.npz (made by tests/golden/make_golden.py from the reference's own code) against the library;
  * oracle/_ref/libfmref.so -- rds-group.cpp compiled in place, ebu-codetables.c pulled in by #include as the reference does -- against
    the fixture and the library, when it is present (this container);
  * prepareText is a member of a class that includes radio.h (the GUI) and cannot be compiled here: its loop is restated below,
    statement by statement, over the REFERENCE's map function and compared with the library's fmx_rds_prepare_text on random buffers
    full of switch pairs."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "ref_rds_tables.npz"))


def golden_names():
    names = bytes(G["pty_utf8"]).split(b"\0")[:64]
    return [[names[2 * p + loc] for loc in range(2)] for p in range(32)]


def test_pty_names_equal_the_reference_table(fmx_amd):
    L = fmx_amd.load_library()
    names = golden_names()
    for p in range(32):
        for loc in range(2):
            assert L.fmx_rds_pty_name(p, loc) == names[p][loc], (p, loc)
    assert names[18][0].decode("utf-8") == "Children’s Progs" and names[0][0] == b"--" and names[21][0] == b"Phone-In"
    assert names[16][1] == b"Rhythm & Blues" and names[18][1] == b"Language"
    for p, loc in ((-1, 0), (32, 0), (0, 2), (0, -1)):
        assert L.fmx_rds_pty_name(p, loc) is None


def test_character_map_equals_the_reference_for_all_256_codes(fmx_amd):
    L = fmx_amd.load_library()
    for alf in range(3):
        got = np.array([L.fmx_rds_map_char(alf, c) for c in range(256)], np.uint16)
        assert (got == G["ebu_map"][alf]).all(), alf
    # the reference's table, not EN 50067's: spot values the clean table would not give
    assert L.fmx_rds_map_char(0, 0x24) == ord("X") and L.fmx_rds_map_char(0, 0x60) == ord(" ") and L.fmx_rds_map_char(0, 0x8D) == 0x3B2
    assert L.fmx_rds_map_char(0, 0x0D) == ord(" ")


def test_reference_build_agrees_with_the_fixture():
    R = ol.ref()
    if R is None:
        pytest.skip("oracle/_ref not built (no reference tree on this box): the fixture stands in")
    names = golden_names()
    for p in range(32):
        for loc in range(2):
            assert R.ref_pty_name(p, loc) == names[p][loc]
    for alf in range(3):
        assert [R.ref_map_ebu(alf, c) for c in range(256)] == G["ebu_map"][alf].tolist()
    for k in range(G["group_blocks"].shape[0]):
        b = np.ascontiguousarray(G["group_blocks"][k]); o = np.zeros(9, np.int32)
        R.ref_rdsgroup_fields(b.ctypes.data_as(C.POINTER(C.c_uint16)), o.ctypes.data_as(C.POINTER(C.c_int32)))
        assert (o == G["group_fields"][k]).all(), k


def test_group_fields_of_the_host_decoder_equal_rdsgroup(fmx_amd):
    """PI, group type, version, PTY out of fmx_rds_decode_bits equal RDSGroup's getters on the same blocks (type A groups: the
    reference does not decode type B ones either)."""
    rng = np.random.default_rng(5)
    blocks, fields = G["group_blocks"], G["group_fields"]
    done = 0
    for k in range(blocks.shape[0]):
        a, b, c, d = (int(v) for v in blocks[k])
        assert fields[k][4] == a and fields[k][5] == (b >> 12) & 15 and fields[k][6] == (b >> 11) & 1 and fields[k][8] == (b >> 5) & 31
        if fields[k][6] or done >= 40:
            continue
        bits = np.array(ol.rds_group_bits(a, b, c, d) * 3, np.uint8)
        info = fmx_amd.fmx.rds_decode_bits(np.concatenate([rng.integers(0, 2, 7).astype(np.uint8), bits]))
        assert info.pi_code == fields[k][4] and info.last_group_type == fields[k][5] and info.pty_code == fields[k][8], k
        done += 1
    assert done == 40


def prepare_text_reference_loop(v, length, alfabet, map_char):
    """rds-groupdecoder.cpp:298-315 + :317-343, one statement per line of the reference."""
    previous = v[0]
    out = []
    i = 1
    while i < length:                                  # for (i = 1; i < length; i++)
        current = v[i]
        if (previous == 0x0F and current == 0x0F) or (previous == 0x0E and current == 0x0E) or (previous == 0x1B and current == 0x6E):
            alfabet = {0x0F: 0, 0x0E: 1, 0x1B: 2}[previous]   # setAlfabetTo (previousChar, currentChar)
            previous = v[i]
            i += 1
        else:
            out.append(map_char(alfabet, previous))
            previous = current
        i += 1
    s = "".join(chr(c) for c in out)
    return s.strip(" "), alfabet                       # QString::trimmed: the map yields no white space but U+0020


def test_prepare_text_with_alphabet_switches(fmx_amd):
    L = fmx_amd.load_library()
    R = ol.ref()
    ref_map = (lambda a, c: R.ref_map_ebu(a, c)) if R is not None else (lambda a, c: int(G["ebu_map"][a][c]))
    rng = np.random.default_rng(3)
    cases = [b"HELLO WORLD     ", b"\x0f\x0fABC", b"AB\x0f\x0fCD  ", b"AB\x1b\x6eCDE", b"  \x0e\x0e\x0e\x0eX ", b"A", b"\x0f", b"caf\x82 \x91\x97 \x0d   ",
             bytes(range(0x20, 0x60)), bytes(range(0xC0, 0x100))]
    for _ in range(300):
        n = int(rng.integers(1, 65))
        buf = rng.integers(0, 256, n).astype(np.uint8)
        for _ in range(int(rng.integers(0, 4))):          # sprinkle switch pairs
            k = int(rng.integers(0, max(n - 1, 1)))
            pair = [(0x0F, 0x0F), (0x0E, 0x0E), (0x1B, 0x6E)][int(rng.integers(0, 3))]
            if k + 1 < n:
                buf[k], buf[k + 1] = pair
        cases.append(bytes(buf))
    for raw in cases:
        v = np.frombuffer(raw + b"\0", np.uint8).copy()
        for length in sorted({len(raw), max(len(raw) - 3, 0), 4 * (len(raw) // 4)}):
            for alf0 in (0, 2):
                exp, alf_exp = prepare_text_reference_loop(v.tolist(), length, alf0, ref_map)
                out = (C.c_uint16 * 80)()
                alf = C.c_uint8(alf0)
                n = L.fmx_rds_prepare_text(v.ctypes.data_as(C.POINTER(C.c_uint8)), length, C.byref(alf), out, 80)
                got = "".join(chr(out[i]) for i in range(n))
                assert got == exp and alf.value == alf_exp, (raw, length, got, exp)


def test_radio_text_through_the_group_decoder(fmx_amd):
    """2A groups whose text holds accented characters and an 0x0F 0x0F pair: fmx_rds_info.radio_text_ucs2 = what the reference's
    Handle_RadioText / prepareText hand to setRadioText (rds-groupdecoder.cpp:222-281, 298-315)."""
    text = b"Caf\x82 \x0f\x0fM\x97nchen \x91 5\xa9"           # (0x82 e-acute, 0x97 u-umlaut, 0x91 a-umlaut, 0xA9 in the reference's table)
    bits = ol.rds_programme_bits(pi=0xD3A1, ps="FMX-AMD ", text=text.decode("latin1"), pty=10)
    info = fmx_amd.fmx.rds_decode_bits(np.concatenate([bits, bits]))
    padded = text + b"\r"
    padded += b" " * (-len(padded) % 4)
    buf = np.frombuffer(padded + b" " * (64 - len(padded)) + b"\0", np.uint8)
    exp, _ = prepare_text_reference_loop(buf.tolist(), 64, 0, lambda a, c: int(G["ebu_map"][a][c]))
    assert info.radio_text_unicode == exp
    # the reference's quirks, spelled out: the switch pair leaves a blank (its second byte) and swallows the 'M' behind it, the last
    # character in front of the padding's end is dropped, 0x97 reads 'ö' and 0xA9 'X' in the reference's table
    assert exp == "Café  önchen ä 5X"
    assert info.pty_code == 10 and fmx_amd.load_library().fmx_rds_pty_name(10, 0) == b"Pop Music"
