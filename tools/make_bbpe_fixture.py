#!/usr/bin/env python3
"""Extract the reference's byte-BPE table -- the 256 integers of PRINTABLE_BASE_CHARS,
K2TransducerAsr/Utils/ByteDataHelper.cs:27-285 -- into tests/golden/bbpe_table.json.

This is the one piece of reference-held DATA on the token -> text path: BYTE_TO_BCHAR[b] = (char)PRINTABLE_BASE_CHARS[b]
(:295-299).  The fixture holds only the integers (data), not the source text.  Run in the build container, where
/root/reference exists; the tests read the committed JSON."""
import json
import os
import re
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/K2TransducerAsr/Utils/ByteDataHelper.cs"
lines = open(ref, encoding="utf-8-sig").read().splitlines()
start = next(i for i, l in enumerate(lines) if "PRINTABLE_BASE_CHARS" in l and "new List<int>" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip() == "};")
vals = [int(m) for l in lines[start + 1 : end] for m in re.findall(r"\b\d+\b", l)]
assert len(vals) == 256, len(vals)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bbpe_table.json")
json.dump({"source": f"K2TransducerAsr/Utils/ByteDataHelper.cs:{start + 1}-{end + 1} (PRINTABLE_BASE_CHARS; BYTE_TO_BCHAR[b] = (char)table[b], :295-299)",
           "bpe_unk": 8263, "bpe_unk_byte": 32, "space_escape": 9601, "table": vals}, open(out, "w"), indent=0)
print("wrote", out, "lines", start + 1, end + 1)
