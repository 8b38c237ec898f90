// C-ABI plumbing (error reporting, version) and the host-side SCST reward scorer.
//
// The scorer restates, in C++ with float64 arithmetic in the reference's order of
// operations, the pure-Python metrics that sit on the SCST training critical path:
//   captionScorer.get_hypo_scores ... common/scst/scorers.py:43-171
//   CiderScorer (CIDEr-D) ........... common/scst/cider_ruotianluo/pyciderevalcap/ciderD/ciderD_scorer.py:130-208
//   BleuScorer ('closest') .......... common/coco_caption/pycocoevalcap/bleu/bleu_scorer.py:60-263
// Hypotheses are scored independently, so they are spread over host threads.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <functional>
#include <thread>
#include <unordered_map>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/comic_hip.h"

#include "common.h"
static thread_local char g_err[512] = "";
thread_local ComicStop g_comic_stop;

void comic_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* comic_last_error(void) { return g_err; }
extern "C" int comic_abi_version(void) { return COMIC_ABI_VERSION; }
extern "C" int comic_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ------------------------------------------------------------------------------ scorer --
namespace {

// N-grams are identified by a 64-bit order-sensitive hash of their words (FNV-1a per word, mixed with the position):
// no strings are allocated per n-gram, the tables are flat vectors sorted by key (sums run in key order:
// deterministic), and the scoring threads stop contending for the allocator.  A collision of two different n-grams
// has probability ~2^-64 per pair.
typedef uint64_t Key;
typedef std::vector<std::pair<Key, int>> Counts;        // sorted by key, keys unique
typedef std::vector<std::pair<Key, double>> Vec;        // sorted by key

inline uint64_t word_hash(const char* p, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) h = (h ^ (unsigned char)p[i]) * 1099511628211ull;
  return h;
}
inline uint64_t mix(uint64_t h, uint64_t w) {
  h ^= w + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
  h *= 0xff51afd7ed558ccdull;
  return h ^ (h >> 33);
}
inline Key ngram_key(const uint64_t* w, int k) {
  uint64_t h = 0x243f6a8885a308d3ull + (uint64_t)k;
  for (int j = 0; j < k; ++j) h = mix(h, w[j]);
  return h;
}

// words of a sentence (split on ASCII white space, as str.split()) as hashes
std::vector<uint64_t> split_ws(const char* s) {
  std::vector<uint64_t> out;
  const char* p = s;
  while (*p) {
    while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\f' || *p == '\v') ++p;
    if (!*p) break;
    const char* q = p;
    while (*q && !(*q == ' ' || *q == '\t' || *q == '\n' || *q == '\r' || *q == '\f' || *q == '\v')) ++q;
    out.push_back(word_hash(p, (size_t)(q - p)));
    p = q;
  }
  return out;
}

// n-gram counts per order
void precook(const std::vector<uint64_t>& w, Counts cnt[4]) {
  for (int k = 1; k <= 4; ++k) {
    Counts& c = cnt[k - 1];
    c.clear();
    std::vector<Key> keys;
    for (size_t i = 0; i + k <= w.size(); ++i) keys.push_back(ngram_key(w.data() + i, k));
    std::sort(keys.begin(), keys.end());
    for (size_t i = 0; i < keys.size();) {
      size_t j = i;
      while (j < keys.size() && keys[j] == keys[i]) ++j;
      c.emplace_back(keys[i], (int)(j - i));
      i = j;
    }
  }
}

struct TfIdf {
  Vec vec[4];
  double norm[4];
  int length;
};

}  // namespace

// Worker threads that live as long as the scorer: creating a thread costs ~100 us on the GPU boxes (containers), which
// was most of a scoring call (16 creations for 256 short hypotheses: 1.8 of 1.9 ms).  run(count, fn) executes fn(0..count-1)
// on the workers and the caller; items are claimed from an atomic counter.
class ScorerPool {
 public:
  explicit ScorerPool(int workers) {
    for (int i = 0; i < workers; ++i) th_.emplace_back([this]() { loop(); });
  }
  ~ScorerPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  int workers() const { return (int)th_.size(); }
  void run(int count, const std::function<void(int)>& fn) {
    if (count <= 0) return;
    if (th_.empty() || count == 1) {
      for (int i = 0; i < count; ++i) fn(i);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn;
      count_ = count;
      left_.store(count);
      next_.store(0);
      ++gen_;
      active_ = true;            // workers join a generation only while it is active (and under this lock)
    }
    cv_.notify_all();
    drain();
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this]() { return left_.load() == 0 && busy_ == 0; });
    active_ = false;             // nobody is inside drain() any more; late wakers keep waiting for the next generation
    fn_ = nullptr;
  }

 private:
  void drain() {
    for (;;) {
      const int i = next_.fetch_add(1);
      if (i >= count_) return;
      (*fn_)(i);
      left_.fetch_sub(1);
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&]() { return stop_ || (active_ && gen_ != seen); });
        if (stop_) return;
        seen = gen_;
        ++busy_;
      }
      drain();
      {
        std::lock_guard<std::mutex> lk(m_);
        --busy_;
      }
      done_.notify_all();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  int count_ = 0, busy_ = 0;
  std::atomic<int> next_{0}, left_{0};
  uint64_t gen_ = 0;
  bool stop_ = false, active_ = false;
};

struct comic_scorer {
  std::unordered_map<Key, double> df;
  double log_ref_len;
  mutable std::mutex pool_mutex;                 // one scoring call at a time uses the pool
  mutable std::unique_ptr<ScorerPool> pool;

  void to_vec(const Counts cnt[4], TfIdf& t) const {
    t.length = 0;
    for (int k = 0; k < 4; ++k) {
      double nrm = 0.0;
      t.vec[k].clear();
      for (const auto& kv : cnt[k]) {
        auto it = df.find(kv.first);
        const double d = log(std::max(1.0, it == df.end() ? 0.0 : it->second));
        const double v = (double)kv.second * (log_ref_len - d);
        t.vec[k].emplace_back(kv.first, v);
        nrm += v * v;
        if (k == 1) t.length += kv.second;  // reference quirk: "length" counts bigrams
      }
      t.norm[k] = sqrt(nrm);
    }
  }

  // a reference sentence cooked once per call: the (1 + beams) hypotheses of an image share their references
  struct CookedRef {
    Counts cnt[4];
    TfIdf tv;
    int words;
  };
  void cook_ref(const char* text, CookedRef& c) const {
    const std::vector<uint64_t> w = split_ws(text);
    c.words = (int)w.size();
    precook(w, c.cnt);
    to_vec(c.cnt, c.tv);
  }

  double cider_one(const Counts hc[4], const std::vector<const CookedRef*>& refs) const {
    TfIdf h;
    to_vec(hc, h);
    double score[4] = {0, 0, 0, 0};
    for (const CookedRef* rp : refs) {
      const TfIdf& rv = rp->tv;
      const double delta = (double)(h.length - rv.length);
      for (int k = 0; k < 4; ++k) {
        double val = 0.0;
        // n-grams present in both (an n-gram missing from the reference contributes min(h, 0) * 0 = 0)
        size_t i = 0, j = 0;
        const Vec &hv = h.vec[k], &rr = rv.vec[k];
        while (i < hv.size() && j < rr.size()) {
          if (hv[i].first < rr[j].first) {
            ++i;
          } else if (rr[j].first < hv[i].first) {
            ++j;
          } else {
            val += std::min(hv[i].second, rr[j].second) * rr[j].second;
            ++i;
            ++j;
          }
        }
        if (h.norm[k] != 0 && rv.norm[k] != 0) val /= (h.norm[k] * rv.norm[k]);
        val *= pow(M_E, -(delta * delta) / (2 * 6.0 * 6.0));
        score[k] += val;
      }
    }
    double avg = (((score[0] + score[1]) + score[2]) + score[3]) / 4.0;
    avg /= (double)refs.size();
    avg *= 10.0;
    return avg;
  }

  static void bleu_one(int testlen, const Counts hc[4], const std::vector<const CookedRef*>& refs, double out[4]) {
    const double small = 1e-9, tiny = 1e-15;
    std::vector<int> reflens;
    for (const CookedRef* rp : refs) reflens.push_back(rp->words);
    // 'closest': min over (|l - testlen|, l)
    int best_d = 1 << 30, reflen = 0;
    for (int l : reflens) {
      const int dd = abs(l - testlen);
      if (dd < best_d || (dd == best_d && l < reflen)) {
        best_d = dd;
        reflen = l;
      }
    }
    double bleu = 1.0;
    for (int k = 0; k < 4; ++k) {
      // clipped matches: for every n-gram of the hypothesis, min(its count, the largest count in any reference)
      int correct = 0;
      for (const auto& kv : hc[k]) {
        int mx = 0;
        for (const CookedRef* rp : refs) {
          const Counts& rc = rp->cnt[k];
          auto it = std::lower_bound(rc.begin(), rc.end(), std::make_pair(kv.first, 0),
                                     [](const std::pair<Key, int>& x, const std::pair<Key, int>& y) { return x.first < y.first; });
          if (it != rc.end() && it->first == kv.first) mx = std::max(mx, it->second);
        }
        correct += std::min(mx, kv.second);
      }
      const int guess = std::max(0, testlen - (k + 1) + 1);
      bleu *= ((double)correct + tiny) / ((double)guess + small);
      out[k] = pow(bleu, 1.0 / (k + 1));
    }
    const double ratio = (testlen + tiny) / (reflen + small);
    if (ratio < 1)
      for (int k = 0; k < 4; ++k) out[k] *= exp(1 - 1 / ratio);
  }
};

extern "C" comic_scorer* comic_scorer_create(const char* ngrams_host, const double* counts_host, int64_t n_entries,
                                             double ref_len) {
  if (!ngrams_host || !counts_host || ref_len <= 0) {
    comic_set_error("scorer_create: bad arguments");
    return nullptr;
  }
  comic_scorer* s = new comic_scorer();
  s->log_ref_len = log(ref_len);
  s->df.reserve((size_t)n_entries * 2);
  const char* p = ngrams_host;
  for (int64_t i = 0; i < n_entries; ++i) {
    const size_t len = strlen(p);
    const std::vector<uint64_t> w = split_ws(p);        // "w1 w2 w3": the words of the n-gram
    if (!w.empty() && w.size() <= 4) s->df.emplace(ngram_key(w.data(), (int)w.size()), counts_host[i]);
    p += len + 1;
  }
  return s;
}

extern "C" void comic_scorer_destroy(comic_scorer* s) { delete s; }

extern "C" int comic_scorer_score(const comic_scorer* s, const char* const* hypos_host, int n,
                                  const char* const* refs_host, const int32_t* refs_per_host, double* out_cider_host,
                                  double* out_bleu_host, int n_threads) {
  if (!s || !hypos_host || !refs_host || !refs_per_host) {
    comic_set_error("scorer_score: null argument");
    return 2;
  }
  std::vector<int64_t> ref_off(n + 1, 0);
  for (int i = 0; i < n; ++i) {
    if (refs_per_host[i] < 1) {
      comic_set_error("scorer_score: hypothesis %d has no reference", i);
      return 2;
    }
    ref_off[i + 1] = ref_off[i] + refs_per_host[i];
  }
  if (n_threads < 1) n_threads = 1;
  n_threads = std::min(n_threads, std::max(1, n));
  // unique reference sentences (by content), cooked once
  const int64_t n_refs = ref_off[n];
  std::unordered_map<std::string, int> uniq;
  std::vector<int> ref_id((size_t)n_refs);
  std::vector<const char*> uniq_text;
  for (int64_t r = 0; r < n_refs; ++r) {
    if (!refs_host[r]) {
      comic_set_error("scorer_score: null reference %lld", (long long)r);
      return 2;
    }
    auto it = uniq.emplace(std::string(refs_host[r]), (int)uniq_text.size());
    if (it.second) uniq_text.push_back(refs_host[r]);
    ref_id[(size_t)r] = it.first->second;
  }
  std::vector<comic_scorer::CookedRef> cooked(uniq_text.size());
  std::lock_guard<std::mutex> pool_lock(s->pool_mutex);
  if (!s->pool || s->pool->workers() != n_threads - 1) s->pool.reset(new ScorerPool(n_threads - 1));
  auto run = [&](int count, const std::function<void(int)>& fn) { s->pool->run(count, fn); };
  run((int)uniq_text.size(), [&](int u) { s->cook_ref(uniq_text[(size_t)u], cooked[(size_t)u]); });
  run(n, [&](int i) {
    const std::vector<uint64_t> hyp = split_ws(hypos_host[i]);
    Counts hc[4];
    precook(hyp, hc);
    std::vector<const comic_scorer::CookedRef*> refs;
    for (int64_t r = ref_off[i]; r < ref_off[i + 1]; ++r) refs.push_back(&cooked[(size_t)ref_id[(size_t)r]]);
    if (out_cider_host) out_cider_host[i] = s->cider_one(hc, refs);
    if (out_bleu_host) comic_scorer::bleu_one((int)hyp.size(), hc, refs, out_bleu_host + (size_t)i * 4);
  });
  // free the cooked references on the workers too: ~100k string-keyed map nodes, 2-3 ms when one thread frees them
  run((int)cooked.size(), [&](int u) { cooked[(size_t)u] = comic_scorer::CookedRef(); });
  return 0;
}


// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78): the checksum of TensorFlow's table blocks
// and tensor-bundle entries (checkpoint container, comic_amd/tf_bundle.py).  Host code; slicing
// by 8 over a table built on first use.
extern "C" uint32_t comic_crc32c(const void* data, size_t n, uint32_t crc) {
  static uint32_t table[8][256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xFF];
    ready = true;
  }
  const unsigned char* p = (const unsigned char*)data;
  uint32_t c = crc ^ 0xFFFFFFFFu;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
        table[3][hi & 0xFF] ^ table[2][(hi >> 8) & 0xFF] ^ table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = table[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}
