// ORBvoc.txt -- the vocabulary in the reference's own on-disk format -- read into the flattened tree the bag-of-words kernels walk
// (orbv_vocab_view, include/orbgpu.h).  Host C++ only.
//
// What is restated: bool TemplatedVocabulary<TDescriptor, F>::loadFromTextFile(const std::string&),
// Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1427, with F = FORB (FORB::fromString, Thirdparty/DBoW2/DBoW2/FORB.cpp:120-135):
//   line 1:   k L scoring weighting        (refused unless 0 <= k <= 20, 1 <= L <= 10, 0 <= scoring <= 5, 0 <= weighting <= 3, :1361-1365)
//   line i:   parent isLeaf d0 .. d31 weight        node id = i - 1 (1-based: node 0 is the root, created before the loop, :1381-1382);
//             the node is appended to its parent's children in file order (:1396); the 32 descriptor bytes are decimal integers
//             cast to unsigned char; weight is read as a double (WordValue); a node with isLeaf > 0 gets the next word id in file
//             order (:1413-1419).  Whether a node IS a leaf when a descriptor walks the tree is decided by children.empty()
//             (Node::isLeaf, :68), not by the flag -- kept.
// The reference's loop is `while(!f.eof()) { getline(f, snode); ... }`: after the last node line the stream is not yet at EOF, so a file
// that ends with a newline (every file saveToTextFile writes: `<< endl`, :1446) yields ONE MORE node from the empty line -- parent 0
// (a failed `>> pid` stores 0), not a leaf, weight 0, and a descriptor nobody initialises (fromString creates the 1 x 32 matrix and
// writes nothing when the extraction fails).  A walk that came closest to that stray child of the root would end there (it has
// no children, so isLeaf() holds) with word 0 / weight 0 and the feature would drop out of the BowVector.  Indeterminate bytes cannot be
// reproduced; the loader's default is the tree the FILE describes (blank trailing lines ignored), and
// ORBV_TEXT_KEEP_TRAILING_NODE appends the stray node with an all-zero descriptor for whoever wants the node COUNT of the reference
// (a zero descriptor is what a fresh heap page gives; the oracle's loader does the same under the same flag).
// Malformed lines -- fewer than 35 fields, a parent that is not an earlier node -- make the reference read garbage; here they are an error.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/orbgpu.h"

struct orbv_text {
  int32_t k = 0, L = 0, scoring = 0, weighting = 0, n_words = 0;
  std::vector<int32_t> child_start, child_ids, word_id, parent;
  std::vector<uint8_t> desc;
  std::vector<double> weight;
};

namespace {

inline const char* skip_ws(const char* p, const char* e) { while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) p++; return p; }
// one decimal integer (optional sign); false at end of line
inline bool take_int(const char*& p, const char* e, long& out) {
  p = skip_ws(p, e);
  if (p >= e) return false;
  bool neg = false;
  if (*p == '-' || *p == '+') { neg = *p == '-'; p++; }
  if (p >= e || *p < '0' || *p > '9') return false;
  long v = 0;
  while (p < e && *p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); p++; }
  out = neg ? -v : v;
  return true;
}
inline bool take_double(const char*& p, const char* e, double& out) {
  p = skip_ws(p, e);
  if (p >= e) return false;
  char buf[64];
  size_t n = 0;
  while (p + n < e && n < sizeof(buf) - 1 && p[n] != ' ' && p[n] != '\t' && p[n] != '\r') { buf[n] = p[n]; n++; }
  buf[n] = 0;
  char* end = nullptr;
  out = std::strtod(buf, &end);           // (what operator>>(double&) accepts for the numbers saveToTextFile writes: %g, 6 significant digits)
  if (end == buf) return false;
  p += (end - buf);
  return true;
}

}  // namespace

extern "C" int orbv_text_load(const char* path, int flags, orbv_text** out) {
  if (!path || !out) return ORBG_BAD_ARG;
  *out = nullptr;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return ORBG_BAD_ARG;
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size <= 0) { close(fd); return ORBG_BAD_ARG; }
  const size_t size = (size_t)st.st_size;
  void* map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (map == MAP_FAILED) return ORBG_INTERNAL;
  const char* const base = static_cast<const char*>(map);
  const char* const end = base + size;
  int rc = ORBG_OK;
  orbv_text* t = new (std::nothrow) orbv_text();
  if (!t) { munmap(map, size); return ORBG_INTERNAL; }
  try {
    // ---- header (:1351-1365)
    const char* p = base;
    const char* eol = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
    if (!eol) eol = end;
    long k = 0, L = 0, n1 = 0, n2 = 0;
    if (!take_int(p, eol, k) || !take_int(p, eol, L) || !take_int(p, eol, n1) || !take_int(p, eol, n2) ||
        k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
      rc = ORBG_BAD_ARG;                     // "Vocabulary loading failure: This is not a correct text file!"
    } else {
      t->k = (int32_t)k; t->L = (int32_t)L; t->scoring = (int32_t)n1; t->weighting = (int32_t)n2;
      // expected_nodes = (k^(L+1) - 1) / (k - 1) (:1372-1374) only sizes a reserve() in the reference; the same here
      double expect = 1; { double pw = 1; for (long l = 0; l < L; l++) { pw *= (double)(k > 1 ? k : 2); expect += pw; } }
      const size_t reserve_n = (size_t)(expect < 4e6 ? expect : 4e6) + 2;
      std::vector<int32_t> parent; parent.reserve(reserve_n);
      std::vector<int32_t> n_children; n_children.reserve(reserve_n);
      t->desc.reserve(reserve_n * 32); t->weight.reserve(reserve_n); t->word_id.reserve(reserve_n);
      parent.push_back(0); n_children.push_back(0);
      t->desc.insert(t->desc.end(), 32, 0); t->weight.push_back(0.0); t->word_id.push_back(0);       // node 0: the root (:1381-1382)
      int32_t n_words = 0;
      p = eol < end ? eol + 1 : end;
      bool stray = false;
      while (p < end && rc == ORBG_OK) {
        eol = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
        if (!eol) eol = end;
        const char* q = skip_ws(p, eol);
        if (q >= eol) {                      // a blank line: the reference makes a node of it (see the header comment)
          stray = true;
          p = eol < end ? eol + 1 : end;
          continue;
        }
        if (stray) { rc = ORBG_BAD_ARG; break; }          // node lines after a blank one: the ids would shift; refuse
        const int32_t nid = (int32_t)parent.size();
        long pid = 0, leaf = 0;
        if (!take_int(q, eol, pid) || !take_int(q, eol, leaf) || pid < 0 || pid >= nid) { rc = ORBG_BAD_ARG; break; }
        uint8_t d[32];
        bool ok = true;
        for (int i = 0; i < 32 && ok; i++) { long v = 0; ok = take_int(q, eol, v); d[i] = (uint8_t)(unsigned char)v; }
        double w = 0;
        if (!ok || !take_double(q, eol, w)) { rc = ORBG_BAD_ARG; break; }
        parent.push_back((int32_t)pid); n_children.push_back(0); n_children[(size_t)pid]++;
        t->desc.insert(t->desc.end(), d, d + 32); t->weight.push_back(w);
        t->word_id.push_back(leaf > 0 ? n_words++ : 0);        // (Node(): word_id(0), TemplatedVocabulary.h:316: what a node keeps when the flag is not set)
        p = eol < end ? eol + 1 : end;
      }
      // (the text after the last newline, if the file does not end with one, was a node line like any other; a file that DOES end with a
      // newline has no further line here -- the reference's extra getline() is what `stray` stands for)
      if (rc == ORBG_OK && size > 0 && base[size - 1] == '\n') stray = true;
      if (rc == ORBG_OK && stray && (flags & ORBV_TEXT_KEEP_TRAILING_NODE)) {
        parent.push_back(0); n_children.push_back(0); n_children[0]++;
        t->desc.insert(t->desc.end(), 32, 0); t->weight.push_back(0.0); t->word_id.push_back(0);
      }
      if (rc == ORBG_OK) {
        const size_t nn = parent.size();
        t->child_start.assign(nn + 1, 0);
        for (size_t i = 0; i < nn; i++) t->child_start[i + 1] = t->child_start[i] + n_children[i];
        t->child_ids.assign((size_t)t->child_start[nn], 0);
        std::vector<int32_t> fill(t->child_start.begin(), t->child_start.end() - 1);
        for (size_t i = 1; i < nn; i++) t->child_ids[(size_t)fill[(size_t)parent[i]]++] = (int32_t)i;      // children in file order (:1396)
        t->n_words = n_words;
        t->parent.swap(parent);
      }
    }
  } catch (const std::bad_alloc&) {
    rc = ORBG_INTERNAL;
  }
  munmap(map, size);
  if (rc != ORBG_OK) { delete t; return rc; }
  *out = t;
  return ORBG_OK;
}

extern "C" int orbv_text_view(const orbv_text* t, orbv_vocab_view* view, int32_t* k, int32_t* scoring, int32_t* n_words) {
  if (!t || !view) return ORBG_BAD_ARG;
  view->n_nodes = (int32_t)t->weight.size();
  view->L = t->L;
  view->weighting = t->weighting;          // WeightingType, Thirdparty/DBoW2/DBoW2/BowVector.h:24-30 = ORBV_TF_IDF ..
  // what the scoring object's mustNormalize() reports (Thirdparty/DBoW2/DBoW2/ScoringObject.h:73-89): L1 for L1_NORM, CHI_SQUARE, KL,
  // BHATTACHARYYA; L2 for L2_NORM; none for DOT_PRODUCT
  view->scoring_norm = t->scoring == 1 ? ORBV_NORM_L2 : t->scoring == 5 ? ORBV_NORM_NONE : ORBV_NORM_L1;
  view->child_start = t->child_start.data();
  view->child_ids = t->child_ids.data();
  view->desc = t->desc.data();
  view->weight = t->weight.data();
  view->word_id = t->word_id.data();
  if (k) *k = t->k;
  if (scoring) *scoring = t->scoring;
  if (n_words) *n_words = t->n_words;
  return ORBG_OK;
}

extern "C" int orbv_text_free(orbv_text* t) {
  delete t;
  return ORBG_OK;
}

extern "C" int orbv_vocab_from_text(int device, const char* path, int flags, orbv_vocab** out) {
  if (!out) return ORBG_BAD_ARG;
  orbv_text* t = nullptr;
  int rc = orbv_text_load(path, flags, &t);
  if (rc != ORBG_OK) return rc;
  orbv_vocab_view v;
  orbv_text_view(t, &v, nullptr, nullptr, nullptr);
  rc = orbv_vocab_create(device, &v, out);
  orbv_text_free(t);
  return rc;
}
