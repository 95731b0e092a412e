"""filesource.py -- the file input of BASELINE configs[0] ("filereader .wav @ 2.304 MS/s"): a deviceHandler-shaped
source over a WAV file, with the semantics of the reference's ``fileHulp`` (devices/filereader/filehulp.cpp:41-147):

* rate and channel count come from the header; 2 channels = (I, Q), 1 channel = I with Q = 0 (:127-137);
* samples are what libsndfile's ``sf_readf_float`` returns: PCM16 / 32768, float32 as is (PCM8, PCM24/32 likewise);
* at end of file the reader seeks back to the start and goes on (:141-143) -- the short read is NOT padded;
* ``getSamples(n)`` hands out n complex samples multiplied by the attenuation factor (:100-119).

No sound library is needed (RIFF/WAVE PCM and IEEE-float parsing only).  ``raw()`` exposes the untouched int16 pairs
of a stereo PCM16 file so that they can go to the GPU as they are (``Fmx.process_host_raw(..., IQ_S16, 32768)``):
4 bytes per complex sample over PCIe instead of 8, bit-identical result.
"""
import struct

import numpy as np


class WavFileSource:
    def __init__(self, path, attenuation=1.0):
        with open(path, "rb") as f:
            data = f.read()
        if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
            raise ValueError("%s: no RIFF/WAVE file" % path)
        pos, fmt, body = 12, None, None
        while pos + 8 <= len(data):
            cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
            if cid == b"fmt ":
                fmt = data[pos + 8:pos + 8 + size]
            elif cid == b"data":
                body = data[pos + 8:pos + 8 + size]
                break
            pos += 8 + size + (size & 1)
        if fmt is None or body is None:
            raise ValueError("%s: fmt or data chunk missing" % path)
        tag, ch, rate, _, _, bits = struct.unpack("<HHIIHH", fmt[:16])
        if tag == 0xFFFE and len(fmt) >= 26:                       # WAVE_FORMAT_EXTENSIBLE: the sub-format's first word
            tag = struct.unpack("<H", fmt[24:26])[0]
        if ch not in (1, 2):
            raise ValueError("%s: %d channels (1 or 2 expected)" % (path, ch))
        self.inputRate, self.numofChannels = rate, ch
        self._s16 = None
        if tag == 1 and bits == 16:
            self._s16 = np.frombuffer(body, "<i2")
            x = self._s16.astype(np.float32) / np.float32(32768.0)
        elif tag == 1 and bits == 8:
            x = (np.frombuffer(body, np.uint8).astype(np.float32) - 128.0) / np.float32(128.0)
        elif tag == 1 and bits == 32:
            x = (np.frombuffer(body, "<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        elif tag == 3 and bits == 32:
            x = np.frombuffer(body, "<f4").astype(np.float32)
        else:
            raise ValueError("%s: format tag %d with %d bits is not supported" % (path, tag, bits))
        frames = len(x) // ch
        x = x[:frames * ch].reshape(frames, ch)
        if ch == 1:
            x = np.stack([x[:, 0], np.zeros(frames, np.float32)], axis=1)
        self._iq = np.ascontiguousarray(x, np.float32)
        self.samplesinFile = frames
        self.currPos = 0
        self.attenuation = float(attenuation)

    # deviceHandler interface (devices/device-handler.h:60-85)
    def getRate(self):
        return self.inputRate

    def Samples(self):
        return 1 << 30                                  # a file never runs dry: it loops

    def _take(self, src, n):
        out, pos = [], self.currPos
        while n > 0:
            k = min(n, self.samplesinFile - pos)
            out.append(src[pos:pos + k])
            pos, n = pos + k, n - k
            if pos >= self.samplesinFile:
                pos = 0                                 # sf_seek (filePointer, 0, SEEK_SET)
        self.currPos = pos
        return np.concatenate(out) if len(out) > 1 else out[0]

    def getSamples(self, n):
        """n complex samples as float32 [n, 2] (I, Q), times the attenuation."""
        v = self._take(self._iq, n)
        return v if self.attenuation == 1.0 else (v * np.float32(self.attenuation)).astype(np.float32)

    def raw(self, n):
        """The same n samples as the file's own int16 pairs [n, 2] (stereo PCM16 files, attenuation 1 only)."""
        if self._s16 is None or self.numofChannels != 2 or self.attenuation != 1.0:
            raise ValueError("raw() needs a stereo PCM16 file and attenuation 1.0")
        return self._take(self._s16.reshape(-1, 2), n)
