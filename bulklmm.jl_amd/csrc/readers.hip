// readers.hip -- the input side of the path: numeric tables from the file formats the reference reads,
// parsed on the host into a column-major table that the caller copies into its own (optionally pinned) buffer.
//   readGenoProb / readGenoProb_ExcludeComplements   src/readData.jl:41-96    CSV, header line, id column, every other column
//   readBXDpheno / readBXDgeno                        src/readData.jl:159-165  CSV, one line skipped, column subsets
//   Helium .he (test/kinship_test.jl:5, Helium.jl)    56-byte header (int64 nrow, int64 ncol, ...) + column-major float64
// Host code only (no HIP call): usable without a GPU.
#include "../../include/bulklmm_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct blmm_table {
  int64_t rows = 0, cols = 0;
  std::vector<double> data;   // column-major
};

extern "C" {

int blmm_read_csv(const char* path, int64_t skip_lines, int64_t first_col, int64_t col_step, int64_t drop_last, blmm_table** out) {
  if (!path || !out || skip_lines < 0 || first_col < 0 || col_step < 1 || drop_last < 0) return BLMM_ERR_INVALID;
  *out = nullptr;
  FILE* f = std::fopen(path, "rb");
  if (!f) return BLMM_ERR_INVALID;
  std::fseek(f, 0, SEEK_END);
  const long sz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::string buf((size_t)(sz > 0 ? sz : 0), '\0');
  if (sz > 0 && std::fread(&buf[0], 1, (size_t)sz, f) != (size_t)sz) { std::fclose(f); return BLMM_ERR_INVALID; }
  std::fclose(f);
  std::vector<double> rowmajor;
  int64_t ncols_sel = -1, nrows = 0;
  size_t pos = 0;
  for (int64_t s = 0; s < skip_lines && pos < buf.size(); ++s) { const size_t e = buf.find('\n', pos); pos = (e == std::string::npos) ? buf.size() : e + 1; }
  std::vector<std::pair<size_t, size_t>> fields;
  while (pos < buf.size()) {
    size_t e = buf.find('\n', pos);
    if (e == std::string::npos) e = buf.size();
    size_t le = e;
    if (le > pos && buf[le - 1] == '\r') --le;
    if (le > pos) {
      fields.clear();
      size_t a = pos;
      bool quoted = false;
      for (size_t i = pos; i <= le; ++i) {
        if (i < le && buf[i] == '"') quoted = !quoted;
        if (i == le || (buf[i] == ',' && !quoted)) { fields.emplace_back(a, i); a = i + 1; }
      }
      const int64_t total = (int64_t)fields.size();
      int64_t cnt = 0;
      for (int64_t c = first_col; c < total - drop_last; c += col_step) {
        size_t fa = fields[(size_t)c].first, fb = fields[(size_t)c].second;
        while (fa < fb && (buf[fa] == ' ' || buf[fa] == '"')) ++fa;
        const char saved = buf[fb];   // strtod needs a terminator: fb <= le <= buf.size(); buf[size()] is the string's '\0'
        if (fb < buf.size()) buf[fb] = '\0';
        char* endp = nullptr;
        const double v = std::strtod(buf.c_str() + fa, &endp);
        if (fb < buf.size()) buf[fb] = saved;
        if (endp == buf.c_str() + fa) return BLMM_ERR_INVALID;   // not a number in a selected column
        rowmajor.push_back(v);
        ++cnt;
      }
      if (ncols_sel < 0) ncols_sel = cnt;
      else if (cnt != ncols_sel) return BLMM_ERR_DIM;
      ++nrows;
    }
    pos = e + 1;
  }
  blmm_table* t = new blmm_table();
  t->rows = nrows; t->cols = ncols_sel < 0 ? 0 : ncols_sel;
  t->data.resize((size_t)t->rows * (size_t)t->cols);
  for (int64_t r = 0; r < t->rows; ++r)
    for (int64_t c = 0; c < t->cols; ++c) t->data[(size_t)c * t->rows + r] = rowmajor[(size_t)r * t->cols + c];
  *out = t;
  return BLMM_OK;
}

int blmm_read_he(const char* path, blmm_table** out) {
  if (!path || !out) return BLMM_ERR_INVALID;
  *out = nullptr;
  FILE* f = std::fopen(path, "rb");
  if (!f) return BLMM_ERR_INVALID;
  int64_t hdr[7];
  if (std::fread(hdr, 8, 7, f) != 7 || hdr[0] < 0 || hdr[1] < 0) { std::fclose(f); return BLMM_ERR_INVALID; }
  blmm_table* t = new blmm_table();
  t->rows = hdr[0]; t->cols = hdr[1];
  t->data.resize((size_t)t->rows * (size_t)t->cols);
  const size_t want = t->data.size();
  if (want && std::fread(t->data.data(), 8, want, f) != want) { std::fclose(f); delete t; return BLMM_ERR_INVALID; }
  std::fclose(f);
  *out = t;
  return BLMM_OK;
}

int64_t blmm_table_rows(const blmm_table* t) { return t ? t->rows : 0; }
int64_t blmm_table_cols(const blmm_table* t) { return t ? t->cols : 0; }
int blmm_table_copy(const blmm_table* t, double* dst) {
  if (!t || (!dst && !t->data.empty())) return BLMM_ERR_INVALID;
  if (!t->data.empty()) std::memcpy(dst, t->data.data(), sizeof(double) * t->data.size());
  return BLMM_OK;
}
void blmm_table_free(blmm_table* t) { delete t; }

}  // extern "C"
