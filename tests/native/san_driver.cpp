// CPU sanitizer driver (AddressSanitizer + UBSan, see `make -C k2transducerasr_amd/csrc san`) for the pure-host units of
// libk2hip that read untrusted bytes: the .k2w container parser and the token -> text stage.  Test infrastructure.
//   san_driver k2w  <file>                 parse once; prints "OK <n_meta> <n_tensors>" or "ERR <code> <message>"
//   san_driver fuzz <file> <seed> <iters>  parse <iters> mutated copies (truncations, byte flips, field overwrites in the header);
//                                          prints the OK / ERR counts.  Any crash or sanitizer report fails the process.
//   san_driver text <tokens.txt> <online 0|1> <id> ...   decode ids to text
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "../../k2transducerasr_amd/csrc/errors.h"
#include "../../k2transducerasr_amd/csrc/k2w_file.h"
#include "../../k2transducerasr_amd/csrc/text.h"

using namespace k2hip;

static int parse(const std::string& path, bool verbose) {
    try {
        K2wFile f(path);
        // touch every tensor's first and last byte: the range checks promised they are inside the mapping
        unsigned long long sum = 0;
        for (const auto& t : f.tensors)
            if (t.nbytes) sum += f.data()[t.off] + f.data()[t.off + t.nbytes - 1];
        if (verbose) printf("OK %zu %zu %llu\n", f.meta.size(), f.tensors.size(), sum);
        return 0;
    } catch (const Error& e) {
        if (verbose) printf("ERR %d %s\n", e.code, e.what());
        return e.code == K2HIP_ERR_IO ? 1 : 2;
    }
}

int main(int argc, char** argv) {
    if (argc >= 3 && !strcmp(argv[1], "k2w")) {
        int r = parse(argv[2], true);
        return r == 2 ? 3 : 0;  // a wrong error class is a failure; OK and K2HIP_ERR_IO are both valid outcomes
    }
    if (argc >= 5 && !strcmp(argv[1], "fuzz")) {
        std::ifstream in(argv[2], std::ios::binary);
        std::vector<char> orig((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (orig.size() < 24) { fprintf(stderr, "cannot read %s\n", argv[2]); return 4; }
        uint64_t data_off = 0;
        memcpy(&data_off, orig.data() + 16, 8);
        const size_t hdr = (size_t)std::min<uint64_t>(data_off, orig.size());
        std::mt19937_64 rng((uint64_t)atoll(argv[3]));
        const int iters = atoi(argv[4]);
        std::string tmp = std::string(argv[2]) + ".fuzz";
        int ok = 0, err = 0, other = 0;
        for (int it = 0; it < iters; it++) {
            std::vector<char> m = orig;
            switch (it % 4) {
                case 0: m.resize(rng() % (it % 8 == 0 ? orig.size() : hdr + 64)); break;          // truncation
                case 1: for (int k = 0; k < 1 + (int)(rng() % 4); k++) m[rng() % hdr] = (char)rng(); break;   // byte flips in the header
                case 2: {   // overwrite a 4- or 8-byte field with an extreme value
                    static const uint64_t ext[] = {0, 1, 0x7fffffffull, 0xffffffffull, 0x7fffffffffffffffull, 0xffffffffffffffffull, 1ull << 40};
                    const uint64_t v = ext[rng() % 7];
                    const size_t w = (rng() & 1) ? 4 : 8, at = rng() % (hdr - w);
                    memcpy(m.data() + at, &v, w);
                    break;
                }
                default: {  // header counts / data offset
                    uint32_t v = (uint32_t)rng();
                    memcpy(m.data() + 8 + 4 * (rng() % 2), &v, 4);
                    if (rng() & 1) { uint64_t d = rng() % (2 * orig.size()); memcpy(m.data() + 16, &d, 8); }
                }
            }
            std::ofstream(tmp, std::ios::binary).write(m.data(), (std::streamsize)m.size());
            int r = parse(tmp, false);
            (r == 0 ? ok : r == 1 ? err : other)++;
        }
        unlink(tmp.c_str());
        printf("fuzz ok=%d err=%d other=%d\n", ok, err, other);
        return other ? 3 : 0;
    }
    if (argc >= 4 && !strcmp(argv[1], "text")) {
        try {
            struct Tab {
                TokenTable* t;
                ~Tab() { token_table_free(t); }
            } tab{token_table_load(argv[2])};
            TokenTable* t = tab.t;
            std::vector<int64_t> ids;
            for (int i = 4; i < argc; i++) ids.push_back(atoll(argv[i]));
            std::string s = decode_tokens(*t, ids.data(), (int)ids.size(), atoi(argv[3]) != 0);
            printf("%s\n", s.c_str());
            for (int b = 0; b < 256; b++)
                if (bbpe_byte_of_char((uint32_t)bbpe_char_of_byte(b)) != b) return 5;
        } catch (const Error& e) {
            printf("ERR %d %s\n", e.code, e.what());
        }
        return 0;
    }
    fprintf(stderr, "usage: san_driver k2w|fuzz|text ...\n");
    return 64;
}
