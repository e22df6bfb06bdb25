// DBoW2 vocabulary file loader and the BowVector / FeatureVector assembly of Frame::ComputeBoW (reference src/Frame.cc:474-481,
// src/System.cc:64-73 -> Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1423 loadFromTextFile, :1126-1208 transform(features, v, fv,
// levelsup), BowVector.cpp:34-85, FeatureVector.cpp:30-44).  Host code only: the tree descent itself is k_bow_transform (match.hip).
#include "common.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

using namespace sslam;

namespace {

// one whitespace-separated integer / real, the way operator>> reads them; false at end of line
inline bool next_long(const char*& p, const char* end, long& v) {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
    if (p >= end) return false;
    char* q = nullptr;
    v = std::strtol(p, &q, 10);
    if (q == p) return false;
    p = q;
    return true;
}
inline bool next_double(const char*& p, const char* end, double& v) {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
    if (p >= end) return false;
    char* q = nullptr;
    v = std::strtod(p, &q);
    if (q == p) return false;
    p = q;
    return true;
}

}  // namespace

extern "C" int sslam_vocab_load_text(sslam_ctx* ctx, const char* path, sslam_vocab** out) {
    if (!ctx || !path || !out) { set_error("sslam_vocab_load_text: invalid arguments"); return SSLAM_ERR_INVALID; }
    FILE* f = std::fopen(path, "rb");
    if (!f) { set_error("sslam_vocab_load_text: cannot open file"); return SSLAM_ERR_INVALID; }
    std::string buf;
    {
        char chunk[1 << 16]; size_t got;
        while ((got = std::fread(chunk, 1, sizeof(chunk), f)) > 0) buf.append(chunk, got);
        std::fclose(f);
    }
    buf.push_back('\n');                                  // strtol/strtod stop at the line end, never past the buffer
    const char* p = buf.data(); const char* fileEnd = buf.data() + buf.size() - 1;
    auto line_end = [&](const char* s) { while (s < fileEnd && *s != '\n') ++s; return s; };
    // header: k L scoring weighting (:1349-1363, same range check)
    const char* e = line_end(p);
    long k = -1, L = -1, n1 = -1, n2 = -1;
    { const char* q = p; if (!next_long(q, e, k) || !next_long(q, e, L) || !next_long(q, e, n1) || !next_long(q, e, n2)) { set_error("sslam_vocab_load_text: not a DBoW2 text vocabulary (header)"); return SSLAM_ERR_INVALID; } }
    if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) { set_error("sslam_vocab_load_text: not a DBoW2 text vocabulary (header range)"); return SSLAM_ERR_INVALID; }
    p = e < fileEnd ? e + 1 : fileEnd;
    // nodes: "parent isLeaf d0 .. d31 weight" per line, node ids in file order starting at 1 (:1376-1419)
    std::vector<int32_t> parent(1, -1), wordId(1, 0), isLeaf(1, 0);
    std::vector<uint8_t> desc(32, 0);
    std::vector<double> weight(1, 0.0);
    int nwords = 0;
    while (p < fileEnd) {
        e = line_end(p);
        const char* q = p;
        long pid;
        if (!next_long(q, e, pid)) { p = e < fileEnd ? e + 1 : fileEnd; continue; }      // blank line (the reference's behaviour there is undefined: INTEGRATION.md)
        const int nid = (int)parent.size();
        long leaf = 0;
        if (pid < 0 || pid >= nid || !next_long(q, e, leaf)) { set_error("sslam_vocab_load_text: malformed node line"); return SSLAM_ERR_INVALID; }
        parent.push_back((int32_t)pid); isLeaf.push_back(leaf > 0);
        desc.resize(desc.size() + 32, 0);
        for (int i = 0; i < 32; ++i) {
            long b;
            if (!next_long(q, e, b)) { set_error("sslam_vocab_load_text: malformed node descriptor"); return SSLAM_ERR_INVALID; }
            desc[(size_t)nid * 32 + i] = (uint8_t)b;
        }
        double w = 0;
        if (!next_double(q, e, w)) { set_error("sslam_vocab_load_text: malformed node weight"); return SSLAM_ERR_INVALID; }
        weight.push_back(w);
        wordId.push_back(leaf > 0 ? nwords : 0);
        if (leaf > 0) ++nwords;
        p = e < fileEnd ? e + 1 : fileEnd;
    }
    const int nnodes = (int)parent.size();
    // children lists in file order (m_nodes[pid].children.push_back(nid)) as CSR
    std::vector<int32_t> childPtr(nnodes + 1, 0), children(std::max(nnodes - 1, 1), 0);
    for (int i = 1; i < nnodes; ++i) ++childPtr[parent[i] + 1];
    for (int i = 0; i < nnodes; ++i) childPtr[i + 1] += childPtr[i];
    { std::vector<int32_t> cur(childPtr.begin(), childPtr.end() - 1); for (int i = 1; i < nnodes; ++i) children[cur[parent[i]]++] = i; }
    for (int i = 1; i < nnodes; ++i) if (isLeaf[i] && childPtr[i + 1] > childPtr[i]) { set_error("sslam_vocab_load_text: a node marked as a word has children"); return SSLAM_ERR_INVALID; }
    int rc = sslam_vocab_create(ctx, nnodes, (int)L, childPtr.data(), children.data(), desc.data(), wordId.data(), weight.data(), out);
    if (rc) return rc;
    (*out)->k = (int)k; (*out)->scoring = (int)n1; (*out)->weighting = (int)n2; (*out)->nwords = nwords;
    return SSLAM_OK;
}

extern "C" int sslam_vocab_set_types(sslam_vocab* v, int weighting, int scoring) {
    if (!v || weighting < 0 || weighting > 3 || scoring < 0 || scoring > 5) { set_error("sslam_vocab_set_types: invalid arguments"); return SSLAM_ERR_INVALID; }
    v->weighting = weighting; v->scoring = scoring;
    return SSLAM_OK;
}

extern "C" int sslam_vocab_info(const sslam_vocab* v, int* k, int* levels, int* scoring, int* weighting, int* nnodes, int* nwords) {
    if (!v) { set_error("sslam_vocab_info: invalid arguments"); return SSLAM_ERR_INVALID; }
    if (k) *k = v->k;
    if (levels) *levels = v->levels;
    if (scoring) *scoring = v->scoring;
    if (weighting) *weighting = v->weighting;
    if (nnodes) *nnodes = v->nnodes;
    if (nwords) *nwords = v->nwords;
    return SSLAM_OK;
}

// BowVector / FeatureVector from the per-feature (word, weight, node) triples, in the reference's feature order
static int assemble_bow(const sslam_vocab* v, const int32_t* word, const double* wt, const int32_t* node, int n, int32_t* bow_word, double* bow_value, int* nbow,
                        int32_t* fv_node, int32_t* fv_ptr, int32_t* fv_feat, int* nfv) {
    std::map<int32_t, double> bow;
    std::map<int32_t, std::vector<int32_t>> fv;
    const bool tf = v->weighting == 0 || v->weighting == 1;                  // TF_IDF, TF: addWeight; IDF, BINARY: addIfNotExist
    for (int i = 0; i < n; ++i) {
        if (!(wt[i] > 0)) continue;                                           // stopped word (:1157)
        auto it = bow.lower_bound(word[i]);
        if (it != bow.end() && it->first == word[i]) { if (tf) it->second += wt[i]; }
        else bow.insert(it, std::make_pair(word[i], wt[i]));
        fv[node[i]].push_back(i);
    }
    const bool must = v->scoring != 5;                                        // every scoring but DOT_PRODUCT normalises (ScoringObject.h:74-89)
    const bool l2 = v->scoring == 1;
    if (tf && !bow.empty() && !must) { const double nd = (double)bow.size(); for (auto& kv : bow) kv.second /= nd; }
    if (must) {
        double norm = 0.0;
        if (!l2) for (auto& kv : bow) norm += std::fabs(kv.second);
        else { for (auto& kv : bow) norm += kv.second * kv.second; norm = std::sqrt(norm); }
        if (norm > 0.0) for (auto& kv : bow) kv.second /= norm;
    }
    int j = 0;
    for (auto& kv : bow) { bow_word[j] = kv.first; bow_value[j] = kv.second; ++j; }
    *nbow = j;
    int a = 0, c = 0;
    fv_ptr[0] = 0;
    for (auto& kv : fv) { fv_node[a] = kv.first; for (int32_t idx : kv.second) fv_feat[c++] = idx; fv_ptr[++a] = c; }
    *nfv = a;
    return SSLAM_OK;
}

extern "C" int sslam_compute_bow(sslam_ctx* ctx, const sslam_vocab* vocab, const uint8_t* desc, int n, int levelsup, int32_t* bow_word, double* bow_value, int* nbow,
                                 int32_t* fv_node, int32_t* fv_ptr, int32_t* fv_feat, int* nfv) {
    if (!ctx || !vocab || n < 0 || !nbow || !nfv || !fv_ptr || (n > 0 && (!desc || !bow_word || !bow_value || !fv_node || !fv_feat))) { set_error("sslam_compute_bow: invalid arguments"); return SSLAM_ERR_INVALID; }
    *nbow = 0; *nfv = 0; fv_ptr[0] = 0;
    if (n == 0) return SSLAM_OK;
    std::vector<int32_t> word(n), node(n); std::vector<double> wt(n);
    int rc = sslam_bow_transform(ctx, vocab, desc, n, levelsup, word.data(), wt.data(), node.data());
    if (rc) return rc;
    return assemble_bow(vocab, word.data(), wt.data(), node.data(), n, bow_word, bow_value, nbow, fv_node, fv_ptr, fv_feat, nfv);
}

extern "C" int sslam_compute_bow_frame(sslam_ctx* ctx, const sslam_vocab* vocab, const sslam_frame* frame, int levelsup, int32_t* bow_word, double* bow_value, int* nbow,
                                       int32_t* fv_node, int32_t* fv_ptr, int32_t* fv_feat, int* nfv) {
    if (!ctx || !vocab || !frame || !nbow || !nfv || !fv_ptr) { set_error("sslam_compute_bow_frame: invalid arguments"); return SSLAM_ERR_INVALID; }
    const int n = sslam_frame_count(frame);
    *nbow = 0; *nfv = 0; fv_ptr[0] = 0;
    if (n == 0) return SSLAM_OK;
    if (!bow_word || !bow_value || !fv_node || !fv_feat) { set_error("sslam_compute_bow_frame: invalid arguments"); return SSLAM_ERR_INVALID; }
    std::vector<int32_t> word(n), node(n); std::vector<double> wt(n);
    int rc = sslam_bow_transform_frame(ctx, vocab, frame, levelsup, word.data(), wt.data(), node.data());
    if (rc) return rc;
    return assemble_bow(vocab, word.data(), wt.data(), node.data(), n, bow_word, bow_value, nbow, fv_node, fv_ptr, fv_feat, nfv);
}
