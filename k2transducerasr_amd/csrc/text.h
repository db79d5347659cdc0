// token ids -> text (text.cpp): DecodeMulti / CheckText / SmartByteDecode of the reference, host only
#pragma once
#include <cstdint>
#include <string>

namespace k2hip {
struct TokenTable;
TokenTable* token_table_load(const char* path);
void token_table_free(TokenTable* t);
int token_table_size(const TokenTable* t);
std::string decode_tokens(const TokenTable& tab, const int64_t* ids, int n, bool online);
}  // namespace k2hip
