// ORACLE (test infrastructure, not product code) -- see orb_oracle.h for status: parity unpinned.
//
// The vocabulary's text format, restated with the stream operations the reference uses:
//   bool TemplatedVocabulary<TDescriptor,F>::loadFromTextFile(const std::string&)   Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1427
//   void TemplatedVocabulary<TDescriptor,F>::saveToTextFile(const std::string&)     :1431-1450
//   FORB::fromString / FORB::toString                                               Thirdparty/DBoW2/DBoW2/FORB.cpp:105-135
// The writer lets a synthetic tree of the reference's size (k = 10, L = 6: ORBvoc.txt has no place in this repository) take the same
// way into the product as the real file: text -> orbv_vocab_from_text.
#include "orb_oracle.h"

#include <cmath>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

struct oracle_vocab {
  int k = 0, L = 0, scoring = 0, weighting = 0, n_words = 0;
  std::vector<std::vector<int32_t>> children;
  std::vector<int32_t> child_start, child_ids, word_id;
  std::vector<uint8_t> desc;
  std::vector<double> weight;
};

// keep_trailing_node != 0: the node the reference's `while(!f.eof())` loop makes of the empty line after the last newline is kept
// (parent 0, no children, weight 0, word id 0; its descriptor -- never written in the reference -- as zeros); 0: blank lines are skipped
extern "C" int oracle_vocab_load_text(const char* path, int keep_trailing_node, oracle_vocab** out) {
  if (!path || !out) return ORBG_BAD_ARG;
  std::ifstream f(path);
  if (!f.is_open()) return ORBG_BAD_ARG;
  oracle_vocab* v = new oracle_vocab();
  std::string s;
  std::getline(f, s);
  std::stringstream ss;
  ss << s;
  int n1 = -1, n2 = -1;
  v->k = -1; v->L = -1;
  ss >> v->k; ss >> v->L; ss >> n1; ss >> n2;
  if (v->k < 0 || v->k > 20 || v->L < 1 || v->L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) { delete v; return ORBG_BAD_ARG; }      // :1361-1365
  v->scoring = n1; v->weighting = n2;
  v->children.resize(1); v->desc.assign(32, 0); v->weight.assign(1, 0.0); v->word_id.assign(1, 0);      // m_nodes.resize(1); m_nodes[0].id = 0
  while (!f.eof()) {
    std::string snode;
    std::getline(f, snode);
    std::stringstream ssnode;
    ssnode << snode;
    if (snode.find_first_not_of(" \t\r") == std::string::npos) {
      if (!keep_trailing_node || !f.eof()) continue;
    }
    const int nid = (int)v->children.size();
    v->children.resize(nid + 1);
    int pid = 0;
    ssnode >> pid;                                   // (a failed extraction stores 0: the stray node hangs under the root)
    if (pid < 0 || pid >= nid) { delete v; return ORBG_BAD_ARG; }
    v->children[pid].push_back(nid);
    int nIsLeaf = 0;
    ssnode >> nIsLeaf;
    uint8_t d[32] = {0};
    for (int iD = 0; iD < 32; iD++) {                // F::L elements through FORB::fromString: int -> unsigned char, skipped on failure
      int n = 0;
      ssnode >> n;
      if (!ssnode.fail()) d[iD] = (unsigned char)n;
    }
    double w = 0;
    ssnode >> w;
    if (ssnode.fail()) { w = 0; if (snode.find_first_not_of(" \t\r") != std::string::npos) { delete v; return ORBG_BAD_ARG; } }
    v->desc.insert(v->desc.end(), d, d + 32);
    v->weight.push_back(w);
    v->word_id.push_back(nIsLeaf > 0 ? v->n_words++ : 0);      // :1413-1419; Node(): word_id(0) otherwise (:316)
  }
  const size_t nn = v->children.size();
  v->child_start.assign(nn + 1, 0);
  for (size_t i = 0; i < nn; i++) { v->child_start[i + 1] = v->child_start[i] + (int32_t)v->children[i].size(); v->child_ids.insert(v->child_ids.end(), v->children[i].begin(), v->children[i].end()); }
  *out = v;
  return ORBG_OK;
}

extern "C" int oracle_vocab_text_view(const oracle_vocab* t, orbv_vocab_view* view, int32_t* k, int32_t* scoring, int32_t* n_words) {
  if (!t || !view) return ORBG_BAD_ARG;
  view->n_nodes = (int32_t)t->weight.size();
  view->L = t->L; view->weighting = t->weighting;
  view->scoring_norm = t->scoring == 1 ? ORBV_NORM_L2 : t->scoring == 5 ? ORBV_NORM_NONE : ORBV_NORM_L1;      // ScoringObject.h:73-89
  view->child_start = t->child_start.data(); view->child_ids = t->child_ids.data(); view->desc = t->desc.data();
  view->weight = t->weight.data(); view->word_id = t->word_id.data();
  if (k) *k = t->k;
  if (scoring) *scoring = t->scoring;
  if (n_words) *n_words = t->n_words;
  return ORBG_OK;
}
extern "C" int oracle_vocab_text_free(oracle_vocab* t) { delete t; return ORBG_OK; }

// saveToTextFile (:1431-1450): "k L  scoring weighting" then, for every node but the root in id order,
// "parent isLeaf d0 d1 .. d31  weight" -- the weight through operator<<(double): six significant digits
extern "C" int oracle_vocab_save_text(const orbv_vocab_view* v, int k, int scoring, const char* path) {
  if (!v || !path) return ORBG_BAD_ARG;
  std::vector<int32_t> parent(v->n_nodes, 0);
  for (int i = 0; i < v->n_nodes; i++)
    for (int c = v->child_start[i]; c < v->child_start[i + 1]; c++) parent[v->child_ids[c]] = i;
  std::fstream f;
  f.open(path, std::ios_base::out);
  if (!f.is_open()) return ORBG_BAD_ARG;
  f << k << " " << v->L << " " << " " << scoring << " " << v->weighting << std::endl;
  std::string line;
  for (int i = 1; i < v->n_nodes; i++) {
    std::stringstream ss;
    ss << parent[i] << " ";
    ss << (v->child_start[i + 1] == v->child_start[i] ? 1 : 0) << " ";
    for (int b = 0; b < 32; b++) ss << (int)v->desc[32 * (size_t)i + b] << " ";        // FORB::toString
    ss << " " << (double)v->weight[i];
    f << ss.str() << "\n";                           // (the reference ends every line with endl: the same bytes, without a flush per node)
  }
  f.close();
  return ORBG_OK;
}
