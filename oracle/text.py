"""CPU restatement of the reference's token -> text stage (TEST INFRASTRUCTURE ONLY, like everything under oracle/).

Follows OfflineRecognizer.DecodeMulti / CheckText / HexToStr (K2TransducerAsr/OfflineRecognizer.cs:432-565), the online
DecodeMulti (OnlineRecognizer.cs:321-352) and Utils/ByteDataHelper.ByteDecode / SmartByteDecode (ByteDataHelper.cs:313-397)
with Python's own regex engine, str methods and UTF-8 codec -- nothing shared with csrc/text.cpp.
PARITY UNPINNED: the reference has no tests for this stage; the known answers in tests/test_text.py are hand-derived.
.NET strings are UTF-16; all strings handled here stay inside the BMP in the tests, so indexes agree.
"""
from __future__ import annotations

import re
import unicodedata

SPACE_ESCAPE = chr(9601)   # ByteDataHelper.cs:23
BPE_UNK = chr(8263)        # :25


def _alphabet():
    """BYTE_TO_BCHAR (ByteDataHelper.cs:27-306): icefall byte_utils -- printable ASCII maps to itself, every other byte to
    the next code point from 256 upward that NFKC normalisation leaves unchanged."""
    b2c, nxt = {}, 256
    for b in range(256):
        if 32 <= b <= 126:
            b2c[b] = chr(b)
            continue
        while unicodedata.normalize("NFKC", chr(nxt)) != chr(nxt):
            nxt += 1
        b2c[b] = chr(nxt)
        nxt += 1
    c2b = {c: b for b, c in b2c.items()}
    c2b[BPE_UNK] = 32
    return b2c, c2b


BYTE_TO_BCHAR, BCHAR_TO_BYTE = _alphabet()


def byte_decode(x: str) -> str:
    try:
        return bytes(BCHAR_TO_BYTE[c] for c in x).decode("utf-8", errors="replace")
    except KeyError:
        return x


def smart_byte_decode(x: str) -> str:
    output = byte_decode(x)
    if output == "":
        n = len(x)
        f = [0] * (n + 1)
        pt = [0] * (n + 1)
        for i in range(1, n + 1):
            f[i] = f[i - 1]
            pt[i] = i - 1
            for j in range(1, min(4, i) + 1):
                if f[i - j] + 1 > f[i] and len(byte_decode(x[i - j : i])) > 0:
                    f[i] = f[i - j] + 1
                    pt[i] = i - j
        cur = n
        while cur > 0:
            if f[cur] == f[pt[cur]] + 1:
                output = byte_decode(x[pt[cur] : cur]) + output
            cur = pt[cur]
    return output


def hex_to_str(hx: str) -> str:
    if len(hx) % 2 != 0:
        hx += "20"
    try:
        data = bytes(int(hx[2 * i : 2 * i + 2], 16) for i in range(len(hx) // 2))  # new byte[hex.Length / 2]
    except ValueError:
        raise ValueError("hex is not a valid hex number!")
    return data.decode("utf-8", errors="replace")


def check_text(text: str) -> str:
    matches = list(re.finditer(r"\<(\w+)\>", text))
    if not matches:
        text = smart_byte_decode(text.replace(" ", ""))
    m_index = -1
    hexs, strs, sb = [], [], ""
    for k, m in enumerate(matches):
        if m_index == -1:
            sb += m.group(0)
        elif m.start() - m_index == 6:
            sb += m.group(0)
        else:
            hexs.append(sb)
            strs.append(sb.replace("<0x", "").replace(">", ""))
            sb = m.group(0)
        if k == len(matches) - 1:
            hexs.append(sb)
            strs.append(sb.replace("<0x", "").replace(">", ""))
        m_index = m.start()
    for h, s in zip(hexs, strs):
        text = text.replace(h, hex_to_str(s))
    return text


def _lower(text: str) -> str:
    # String.ToLower: one-to-one case mapping (Python's str.lower() expands a few code points, e.g. U+0130)
    return "".join(c.lower() if len(c.lower()) == 1 else c for c in text)


def decode_tokens(token_lines, ids, online: bool = False) -> str:
    text = ""
    for t in ids:
        if t == 2:
            break
        if t == -1 and not online:
            continue
        sym = token_lines[t].split(" ")[0]
        if sym not in ("<blk>", "<sos/eos>", "<unk>"):
            text += sym
    return _lower(check_text(text.replace(SPACE_ESCAPE, " ")))


def read_tokens(path: str):
    with open(path, "rb") as f:
        data = f.read()
    if data.startswith(b"\xef\xbb\xbf"):
        data = data[3:]
    return data.decode("utf-8").splitlines()
