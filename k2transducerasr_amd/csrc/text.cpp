// Token ids -> text: the host stage after the search.
//
// Restates OfflineRecognizer.DecodeMulti / CheckText / HexToStr (OfflineRecognizer.cs:432-565), the online twin
// (OnlineRecognizer.cs:321-352) and Utils/ByteDataHelper.ByteDecode / SmartByteDecode (ByteDataHelper.cs:313-397).
// Pure host code over UTF-8 strings; .NET strings are UTF-16, so "6 chars apart" below is counted in UTF-16 units.
//
//   for id in Tokens:  id == 2 -> stop;  id == -1 -> skip (offline only);  sym = tokens.txt[id].Split(' ')[0];
//                      append unless sym in {<blk>, <sos/eos>, <unk>}
//   text = text.Replace("▁", " ")
//   CheckText: matches of \<(\w+)\>
//     none  -> text = SmartByteDecode(text.Replace(" ", ""))      (every space is dropped, then the byte-BPE decode)
//     some  -> runs of matches whose start indexes are exactly 6 apart ("<0xE4><0xBD><0xA0>") are concatenated, each run is
//              hex-decoded to UTF-8 and replaces its text (String.Replace: every occurrence, in run order)
//   ToLower
// Byte-BPE alphabet (icefall byte_utils.py, from fairseq): byte b -> chr(b) for 32..126, else the next code point from 256
// upward that is unchanged by NFKC normalisation (306, 307, 319, 320, 329, 383 are skipped); unknown char 8263 -> byte 32.
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "errors.h"
#include "text.h"

namespace k2hip {

namespace {

// ---- UTF-8 <-> code points (WHATWG decoder: every maximal invalid subpart becomes U+FFFD, as Encoding.UTF8.GetString does)
std::vector<uint32_t> utf8_decode(const std::string& s) {
    std::vector<uint32_t> out;
    size_t i = 0, n = s.size();
    while (i < n) {
        unsigned char c = (unsigned char)s[i];
        if (c < 0x80) { out.push_back(c); i++; continue; }
        int need = 0;
        uint32_t cp = 0;
        unsigned char lo = 0x80, hi = 0xBF;
        if (c >= 0xC2 && c <= 0xDF) { need = 1; cp = c & 0x1F; }
        else if (c >= 0xE0 && c <= 0xEF) { need = 2; cp = c & 0x0F; if (c == 0xE0) lo = 0xA0; if (c == 0xED) hi = 0x9F; }
        else if (c >= 0xF0 && c <= 0xF4) { need = 3; cp = c & 0x07; if (c == 0xF0) lo = 0x90; if (c == 0xF4) hi = 0x8F; }
        else { out.push_back(0xFFFD); i++; continue; }
        size_t j = i + 1;
        bool ok = true;
        for (int k = 0; k < need; k++, j++) {
            if (j >= n) { ok = false; break; }
            unsigned char d = (unsigned char)s[j];
            if (d < lo || d > hi) { ok = false; break; }
            cp = (cp << 6) | (d & 0x3F);
            lo = 0x80; hi = 0xBF;
        }
        if (ok) { out.push_back(cp); i = j; }
        else { out.push_back(0xFFFD); i = j; }  // the bytes consumed so far form one maximal subpart
    }
    return out;
}
void utf8_append(std::string& s, uint32_t cp) {
    if (cp < 0x80) s.push_back((char)cp);
    else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    else { s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
}
std::string utf8_encode(const std::vector<uint32_t>& v) {
    std::string s;
    for (uint32_t c : v) utf8_append(s, c);
    return s;
}
int utf16_units(uint32_t cp) { return cp >= 0x10000 ? 2 : 1; }

// \w of the regexes: letters / digits / underscore.  ASCII exactly; beyond ASCII every code point except the
// general-punctuation, symbol and CJK-punctuation blocks is taken as a letter (a simplification of Unicode's L/Mn/Nd/Pc
// classes that is exact for the alphabets the reference's model zoo uses: Latin, CJK ideographs, kana, hangul).
bool is_word(uint32_t c) {
    if (c < 0x80) return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '_';
    if (c >= 0x80 && c <= 0xBF) return c == 0xAA || c == 0xB5 || c == 0xBA;
    if (c == 0xD7 || c == 0xF7) return false;
    if (c >= 0x2000 && c <= 0x2BFF) return false;   // punctuation, arrows, math, box drawing (incl. U+2581, U+2047)
    if (c >= 0x3000 && c <= 0x303F) return c == 0x3005 || c == 0x3006 || c == 0x3007;
    if (c >= 0xFF00 && c <= 0xFF0F) return false;
    if (c == 0xFFFD) return false;
    return true;
}

// simple (one-to-one) lower-casing of the scripts with case that the model zoo's vocabularies contain
uint32_t to_lower(uint32_t c) {
    if (c >= 'A' && c <= 'Z') return c + 32;
    if (c < 0x80) return c;
    if (c >= 0xC0 && c <= 0xDE && c != 0xD7) return c + 32;                                 // Latin-1
    if (c >= 0x100 && c <= 0x137) return (c & 1) ? c : c + 1;                                // Latin Extended-A pairs
    if (c >= 0x139 && c <= 0x148) return (c & 1) ? c + 1 : c;
    if (c >= 0x14A && c <= 0x177) return (c & 1) ? c : c + 1;
    if (c == 0x178) return 0xFF;
    if (c >= 0x179 && c <= 0x17E) return (c & 1) ? c + 1 : c;
    if (c >= 0x391 && c <= 0x3A9 && c != 0x3A2) return c + 32;                               // Greek
    if (c >= 0x410 && c <= 0x42F) return c + 32;                                             // Cyrillic
    if (c >= 0x400 && c <= 0x40F) return c + 80;
    if (c >= 0xFF21 && c <= 0xFF3A) return c + 32;                                           // full-width Latin
    return c;
}

struct ByteAlphabet {
    uint32_t b2c[256];
    ByteAlphabet() {
        static const uint32_t skip[] = {306, 307, 319, 320, 329, 383};  // changed by NFKC
        uint32_t next = 256;
        for (int b = 0; b < 256; b++) {
            if (b >= 32 && b <= 126) { b2c[b] = (uint32_t)b; continue; }
            for (;;) {
                bool sk = false;
                for (uint32_t s : skip) sk |= (s == next);
                if (!sk) break;
                next++;
            }
            b2c[b] = next++;
        }
    }
    int byte_of(uint32_t c) const {
        if (c == 8263) return 32;  // BPE_UNK -> space (ByteDataHelper.cs:304)
        for (int b = 0; b < 256; b++)
            if (b2c[b] == c) return b;
        return -1;
    }
};
const ByteAlphabet& alphabet() {
    static const ByteAlphabet a;
    return a;
}

// ByteDataHelper.ByteDecode (:330-345): a char outside the alphabet throws inside the try -> the input comes back unchanged
std::string byte_decode(const std::vector<uint32_t>& x) {
    std::string bytes;
    for (uint32_t c : x) {
        if (c >= 0x10000) return utf8_encode(x);  // a surrogate half is not in the alphabet either
        int b = alphabet().byte_of(c);
        if (b < 0) return utf8_encode(x);
        bytes.push_back((char)b);
    }
    return utf8_encode(utf8_decode(bytes));  // Encoding.UTF8.GetString: invalid sequences -> U+FFFD
}

// ByteDataHelper.SmartByteDecode (:352-397): the DP only runs when the plain decode is empty
std::string smart_byte_decode(const std::vector<uint32_t>& x) {
    std::string out = byte_decode(x);
    if (!out.empty()) return out;
    const int n = (int)x.size();
    std::vector<int> f(n + 1, 0), pt(n + 1, 0);
    for (int i = 1; i <= n; i++) {
        f[i] = f[i - 1];
        pt[i] = i - 1;
        for (int j = 1; j <= std::min(4, i); j++) {
            std::vector<uint32_t> sub(x.begin() + (i - j), x.begin() + i);
            if (f[i - j] + 1 > f[i] && !byte_decode(sub).empty()) { f[i] = f[i - j] + 1; pt[i] = i - j; }
        }
    }
    for (int cur = n; cur > 0; cur = pt[cur])
        if (f[cur] == f[pt[cur]] + 1) out = byte_decode(std::vector<uint32_t>(x.begin() + pt[cur], x.begin() + cur)) + out;
    return out;
}

void replace_all(std::string& s, const std::string& from, const std::string& to) {
    if (from.empty()) return;
    size_t pos = 0;
    while ((pos = s.find(from, pos)) != std::string::npos) {
        s.replace(pos, from.size(), to);
        pos += to.size();
    }
}

// HexToStr (:537-565)
std::string hex_to_str(std::string hex) {
    if (hex.size() % 2 != 0) hex += "20";
    std::string bytes;
    for (size_t i = 0; i + 1 < hex.size(); i += 2) {
        int v = 0;
        for (int k = 0; k < 2; k++) {
            char ch = hex[i + k];
            int d = (ch >= '0' && ch <= '9') ? ch - '0' : (ch >= 'a' && ch <= 'f') ? ch - 'a' + 10 : (ch >= 'A' && ch <= 'F') ? ch - 'A' + 10 : -1;
            if (d < 0) failf(K2HIP_ERR_INVALID, "hex is not a valid hex number!");  // the reference throws ArgumentException
            v = v * 16 + d;
        }
        bytes.push_back((char)v);
    }
    return utf8_encode(utf8_decode(bytes));
}

// CheckText (:478-535)
std::string check_text(const std::string& text_in) {
    std::vector<uint32_t> cp = utf8_decode(text_in);
    // matches of \<(\w+)\> with their UTF-16 start index and UTF-8 text
    struct Match { long idx16; std::string s; };
    std::vector<Match> ms;
    {
        long idx16 = 0;
        size_t i = 0;
        std::vector<long> pos16(cp.size() + 1, 0);
        for (size_t k = 0; k < cp.size(); k++) pos16[k + 1] = pos16[k] + utf16_units(cp[k]);
        (void)idx16;
        while (i < cp.size()) {
            if (cp[i] == '<') {
                size_t j = i + 1;
                while (j < cp.size() && is_word(cp[j])) j++;
                if (j > i + 1 && j < cp.size() && cp[j] == '>') {
                    ms.push_back({pos16[i], utf8_encode(std::vector<uint32_t>(cp.begin() + i, cp.begin() + j + 1))});
                    i = j + 1;
                    continue;
                }
            }
            i++;
        }
    }
    std::string text = text_in;
    if (ms.empty()) {
        std::vector<uint32_t> nosp;
        for (uint32_t c : cp)
            if (c != ' ') nosp.push_back(c);
        return smart_byte_decode(nosp);
    }
    std::vector<std::string> hexs, strs;
    std::string run;
    long m_index = -1;
    auto flush = [&]() {
        hexs.push_back(run);
        std::string st = run;
        replace_all(st, "<0x", "");
        replace_all(st, ">", "");
        strs.push_back(st);
    };
    for (size_t k = 0; k < ms.size(); k++) {
        if (m_index == -1) run += ms[k].s;
        else if (ms[k].idx16 - m_index == 6) run += ms[k].s;
        else { flush(); run = ms[k].s; }
        if (k + 1 == ms.size()) flush();
        m_index = ms[k].idx16;
    }
    for (size_t k = 0; k < hexs.size(); k++) replace_all(text, hexs[k], hex_to_str(strs[k]));
    return text;
}

}  // namespace

struct TokenTable {
    std::vector<std::string> lines;  // File.ReadAllLines (OfflineRecognizer.cs:36)
};

TokenTable* token_table_load(const char* path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) failf(K2HIP_ERR_IO, "cannot open tokens file %s", path);
    std::string all((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (all.size() >= 3 && (unsigned char)all[0] == 0xEF && (unsigned char)all[1] == 0xBB && (unsigned char)all[2] == 0xBF) all.erase(0, 3);
    auto* t = new TokenTable();
    size_t i = 0;
    while (i < all.size()) {  // line terminators: \n, \r, \r\n; no empty last line
        size_t j = i;
        while (j < all.size() && all[j] != '\n' && all[j] != '\r') j++;
        t->lines.emplace_back(all, i, j - i);
        if (j < all.size() && all[j] == '\r' && j + 1 < all.size() && all[j + 1] == '\n') j++;
        i = j + 1;
    }
    return t;
}
void token_table_free(TokenTable* t) { delete t; }
int token_table_size(const TokenTable* t) { return (int)t->lines.size(); }

// DecodeMulti for one stream; online = OnlineRecognizer's variant (no special case for id -1)
std::string decode_tokens(const TokenTable& tab, const int64_t* ids, int n, bool online) {
    std::string text;
    for (int i = 0; i < n; i++) {
        const int64_t id = ids[i];
        if (id == 2) break;
        if (id == -1 && !online) continue;
        if (id < 0 || id >= (int64_t)tab.lines.size())
            failf(K2HIP_ERR_INVALID, "token id %lld outside tokens.txt (%zu lines)", (long long)id, tab.lines.size());
        const std::string& line = tab.lines[(size_t)id];
        std::string sym = line.substr(0, line.find(' '));
        if (sym != "<blk>" && sym != "<sos/eos>" && sym != "<unk>") text += sym;
    }
    replace_all(text, "\xE2\x96\x81", " ");  // U+2581
    std::string checked = check_text(text);
    std::vector<uint32_t> cp = utf8_decode(checked);
    for (uint32_t& c : cp) c = to_lower(c);
    return utf8_encode(cp);
}

// BYTE_TO_BCHAR / BCHAR_TO_BYTE of the reference (ByteDataHelper.cs:27-306) as this file generates them; pinned against the
// reference's own 256-entry table by tests/test_text.py::test_bbpe_alphabet_equals_reference_table
int bbpe_char_of_byte(int b) { return (b < 0 || b > 255) ? -1 : (int)alphabet().b2c[b]; }
int bbpe_byte_of_char(uint32_t cp) { return alphabet().byte_of(cp); }

}  // namespace k2hip
