// token ids -> text (text.cpp): DecodeMulti / CheckText / SmartByteDecode of the reference, host only
#pragma once
#include <cstdint>
#include <string>

namespace k2hip {
struct TokenTable;
TokenTable* token_table_load(const char* path);
void token_table_free(TokenTable* t);
int token_table_size(const TokenTable* t);
int bbpe_char_of_byte(int b);          // code point of BYTE_TO_BCHAR[b], -1 outside 0..255
int bbpe_byte_of_char(uint32_t cp);    // BCHAR_TO_BYTE[cp] (BPE_UNK 8263 -> 32), -1 if cp is not in the alphabet
std::string decode_tokens(const TokenTable& tab, const int64_t* ids, int n, bool online);
}  // namespace k2hip
