// Host side of the `sedef align generate` stage above the DP C ABI (include/sedef_hip.h).
//
// Restates, in a batched form, the reference's per-pair logic around the DP kernel:
//   Alignment (src/align.{h,cc}), Hit / BED I/O (src/hit.{h,cc}), FASTA access (src/fasta.{h,cc}),
//   seed chaining (src/chain.cc, src/segment.{h,tpp}), chain refinement (src/refine.cc) and the
//   stage driver (src/align_main.cc:285-337).
// Every DP the reference runs inline through align_helper (src/align.cc:39-68) is collected here into
// batches for the GPU: the per-pair code is run once in "recording" mode to enumerate its DP requests
// (their inputs never depend on earlier DP results inside one phase), the batch goes to
// sdf_extz2_batch, and the same code is run again with the results.
#pragma once
#include <cstdint>
#include <deque>
#include <functional>
#include <algorithm>
#include <initializer_list>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace sdfh {

// CIGAR runs (op, length).  The reference keeps them in a std::deque; libstdc++'s deque allocates two blocks even
// when empty and the stage builds millions of short ones, so this is a vector with a movable front instead (the
// operations the alignment code uses: both ends, range insert at either end, random-access iteration).
class Cigar {
 public:
  typedef std::pair<char, int> value_type;
  typedef value_type *iterator;
  typedef const value_type *const_iterator;
  // number of match columns ('|' of the reference's alignment string) among the M runs, when the producer knows it
  // (the device's per-task counter, summed over a request's tasks); -1: unknown, counted on the host when needed
  int matches = -1;
  Cigar() {}
  Cigar(std::initializer_list<value_type> il) : v_(il) {}
  iterator begin() { return v_.data() + head_; }
  iterator end() { return v_.data() + v_.size(); }
  const_iterator begin() const { return v_.data() + head_; }
  const_iterator end() const { return v_.data() + v_.size(); }
  size_t size() const { return v_.size() - head_; }
  bool empty() const { return v_.size() == head_; }
  void clear() {
    v_.clear();
    head_ = 0;
  }
  value_type &operator[](size_t i) { return v_[head_ + i]; }
  const value_type &operator[](size_t i) const { return v_[head_ + i]; }
  value_type &front() { return v_[head_]; }
  const value_type &front() const { return v_[head_]; }
  value_type &back() { return v_.back(); }
  const value_type &back() const { return v_.back(); }
  void push_back(const value_type &x) { v_.push_back(x); }
  void pop_back() { v_.pop_back(); }
  void pop_front() { ++head_; }
  void push_front(const value_type &x) { insert(begin(), &x, &x + 1); }
  // `at` is begin() or end() (all the alignment code needs)
  void insert(const_iterator at, const_iterator first, const_iterator last) {
    const size_t n = (size_t)(last - first);
    if (n == 0) return;
    if (at == end() && !(at == begin() && head_ >= n)) {
      v_.insert(v_.end(), first, last);
    } else if (head_ >= n) {
      head_ -= n;
      std::copy(first, last, v_.data() + head_);
    } else {
      v_.insert(v_.begin() + (long)head_, first, last);
    }
  }

 private:
  std::vector<value_type> v_;
  size_t head_ = 0;
};

// ---- tunables (reference: src/globals.{h,cc}) ------------------------------------------------
struct Params {
  int match = 5, mismatch = -4, gap_open = -40, gap_extend = -1;  // Align::* (globals.cc:25-28)
  int max_ksw_seq_len = 60 * 1000;                                 // Align::MAX_KSW_SEQ_LEN
  int kmer = 11;
  // Chain::* / Chain::Refine::* / Search::* values the stage reads (globals.h:71-87, globals.cc:20-30)
  int min_uppercase_match = 90, match_chain_score = 4, max_chain_gap = 210;
  double refine_match = 10, refine_mismatch = 1, refine_gap = 0.5, refine_gapopen = 100;
  int refine_min_read = 900, refine_side_align = 500, refine_max_gap = 10 * 1000;
  int min_read_size = 700;
  double max_error = 0.30;
};

struct Anchor {  // reference: src/align.h:25-28
  int q, r, l;
  int has_u;
};

// ---- DP requests ---------------------------------------------------------------------------------
// A sequence the stage only points to: a candidate pair's query or reference as the driver fetched it -- since round 6
// straight into the provider's pinned character pool, where the anchors call uploads it from and the DP rounds' requests
// point into (the reference holds a std::string per pair and cuts substrings out of it, src/align_main.cc:303-306).
struct SeqView {
  const char *p = nullptr;
  size_t n = 0;
  SeqView() {}
  SeqView(const char *p_, size_t n_) : p(p_), n(n_) {}
  SeqView(const std::string &s) : p(s.data()), n(s.size()) {}  // (the string must outlive the view)
  const char *data() const { return p; }
  size_t size() const { return n; }
  const char *begin() const { return p; }
  const char *end() const { return p + n; }
  char operator[](size_t i) const { return p[i]; }
  std::string str() const { return std::string(p, n); }
};

// One align_helper call (src/align.cc:39-68) before align_dna: two ranges of raw FASTA characters.  The ranges point into
// the pair's own two sequences -- the reference cuts substrings out of them (src/align.cc:129-175,235-242,583-590); here
// nothing is copied: a provider that keeps the super-batch's characters in HBM turns the pointers into offsets of that pool
// (ResidentReq), the others code the bases where they build their task pool.  The sequences must outlive the request.
struct DpRequest {
  const char *q, *t;
  int qlen, tlen;
};

// Runs a batch of align_helper-equivalent requests; returns one CIGAR (ops M/D/I) per request.
class DpProvider {
 public:
  virtual ~DpProvider() {}
  virtual std::vector<Cigar> run(const std::vector<DpRequest> &reqs, const Params &p) = 0;
  // Optional raw form: the device's CIGAR words of every DP task, and the tasks of every request; the caller turns
  // them into Cigars where it consumes them (the stage driver does that on the thread that owns the pair, so that
  // the memory of a pair is allocated and freed by one thread).  Returns false if the provider has no raw form.
  struct Raw {
    struct Rec {       // per task (the layout of sdf_result_brief)
      int64_t off;     // first word
      int32_t cnt;     // number of words
      int32_t match;   // match columns (sdf_result.matches)
    };
    // records and words either in memory of this object or where the provider's device copies landed (its pinned
    // staging: valid until the provider's next DP call -- the driver has consumed a round's results by then)
    const Rec *recs = nullptr;
    const uint32_t *words = nullptr;
    std::unique_ptr<Rec[]> own_recs;
    std::unique_ptr<uint32_t[]> own_words;
    std::vector<size_t> first_task;  // per request (+1): its tasks are [first_task[r], first_task[r+1])
    Cigar cigar(size_t req) const;
  };
  virtual bool run_raw(const std::vector<DpRequest> &, const Params &, Raw &) { return false; }
  // Optional: the same for requests named as ranges of the character pool the provider's last anchors() call left on the
  // device (AnchorBatch::resident; offsets = q_base / r_base of the pair + the range's start in the pair's sequence).
  struct ResidentReq {
    int64_t q_off, t_off;
    int32_t qlen, tlen;
  };
  virtual bool run_resident(const std::vector<ResidentReq> &, const Params &, Raw &) { return false; }
  // Optional: another provider of the same kind (own device context) for a second lane of the stage driver.
  // (device < 0: the same device as this one)
  virtual std::unique_ptr<DpProvider> clone(int /*device*/ = -1) { return nullptr; }
  // ... and back again: a lane that is done with a clone()d provider returns it, so that the next run of the stage in this
  // process (several buckets, one process: generate_many) takes it again instead of setting up another device context.
  // The default lets it go.
  virtual void give_back(std::unique_ptr<DpProvider> p) { p.reset(); }
  // Optional: the bytes of sequence the largest super-batch of the run holds, told before the provider's first call: a
  // provider with device and pinned buffers sizes them once, on a thread of its own, while the driver fetches sequences.
  virtual void prepare(size_t /*max_batch_bytes*/) {}
  // Optional: generate_anchors for a batch of pairs on the device.  Returns false if the provider cannot do it
  // for these inputs (the caller then computes them on the host with generate_anchors()).
  struct AnchorJob {
    SeqView query, ref;
    bool same_chr;
    int delta;
  };
  // Optional: a host buffer of `bytes` the provider would like the super-batch's sequences fetched INTO (pinned memory it
  // uploads from: AnchorJob views inside it are not copied again).  Valid until the provider's next pool_host() call.
  // NULL: the driver keeps the sequences in memory of its own.
  virtual char *pool_host(size_t /*bytes*/) { return nullptr; }
  // The anchors of pair k are flat.buf[flat.off[k] .. flat.off[k+1]).
  struct AnchorBatch {
    std::unique_ptr<Anchor[]> buf;
    const Anchor *view = nullptr;  // ... or the provider's pinned staging (valid until its next anchors() call)
    const Anchor *data() const { return view ? view : buf.get(); }
    std::vector<int64_t> off;
    // resident: the pairs' characters stay on the device until the provider's next anchors() call -- pair k's query at
    // q_base[k], its reference at r_base[k] of that pool -- and run_resident() takes requests in those coordinates
    bool resident = false;
    std::vector<int64_t> q_base, r_base;
  };
  virtual bool anchors(const std::vector<AnchorJob> &, int /*kmer*/, AnchorBatch &) { return false; }
  // Optional: the anchors of MORE pairs of the same pool while the caller is still reading the `keep` anchors the call
  // before returned (they stay valid).  false: not possible right now -- the caller calls anchors() when it is done reading.
  virtual bool anchors_more(const std::vector<AnchorJob> &, int /*kmer*/, size_t /*keep*/, AnchorBatch &) { return false; }
  // Optional: the first `bytes` of the pool_host() buffer are complete (the provider may send them to the device now).
  virtual void pool_ready(size_t /*bytes*/) {}
  int64_t tasks = 0, cells = 0;  // statistics
  double t_pack = 0, t_call = 0, t_unpack = 0;  // wall seconds inside run(): request packing, device call, unpacking
};

// The product provider: sdf_extz2_batch on a HIP device.  Throws std::string when no device / library.
std::unique_ptr<DpProvider> make_gpu_provider(int device);
// The same with `lanes` - 1 more providers created side by side with the first (threads of their own; a device context
// costs 50-120 ms, most of it its four streams): clone() of the returned provider hands them out.  `devices`: the
// ordinals the lanes after the first go round-robin over (empty: all on `device`).
// `max_batch_bytes` (0: unknown): as DpProvider::prepare takes it -- every provider then sizes its buffers where it is set up.
std::unique_ptr<DpProvider> make_gpu_providers(int device, int lanes, const std::vector<int> &devices, size_t max_batch_bytes);
// What the stage driver will do with a BED file of seed pairs, from a light pass over its lines: pairs, lanes (and the
// devices of SDF_DEVICES), pairs per super-batch, an upper bound of the sequence bytes of the largest super-batch.
struct StageHint {
  int pairs = 0, lanes = 1, super_batch = 8192;
  std::vector<int> devices;
  size_t max_batch_bytes = 0;
};
StageHint stage_hint(const std::string &bed_path, int super_batch = 8192);
// Lanes of the stage driver for `total` seed pairs (SDF_LANES / SDF_DEVICES in the environment decide otherwise), and the
// device list of SDF_DEVICES.
int stage_lane_count(int total, std::vector<int> *devices);

// TEST HOOK: a provider that calls a single-task function with the oracle's signature
// (oracle/extz2_oracle.h: sdfo_extz2).  Never constructed by the CLI.
typedef void (*test_dp_fn)(int, const uint8_t *, int, const uint8_t *, int, const int8_t *, int, int, int, int,
                           int, void *);
std::unique_ptr<DpProvider> make_test_provider(test_dp_fn fn);

// DP access handed to the restated reference code: records requests or replays results.
struct DpSession {
  bool recording = true;
  std::vector<DpRequest> *requests = nullptr;  // recording: appended to
  const std::vector<Cigar> *results = nullptr; // replay: consumed in order
  size_t cursor = 0;
  // true: both sequences of the pair hold nothing but ACGTN (either case), so a match on the DP's codes is a match of
  // the reference's character comparison (src/align.cc:29-35) and the device's match counter can be taken as it is
  bool codes_are_exact = false;
  // align_helper on two ranges of raw FASTA characters (align_dna is the provider's business)
  Cigar align_ranges(const char *q, int qlen, const char *t, int tlen);
};

// ---- sequences / hits ----------------------------------------------------------------------------
struct Sequence {  // reference: src/hash.h:51-57, src/hash.cc:104-109
  std::string name, seq;
  bool is_rc;
  Sequence(const std::string &name, const std::string &seq, bool is_rc = false);
};

struct Hit;

// The alignment of a query range [start_a, end_a) with a reference range [start_b, end_b) of one candidate pair
// (reference: class Alignment, src/align.h:32-103, src/align.cc).
//
// The reference object owns copies of both substrings and three per-column strings (align_a / align_b / alignment) that
// every operation rebuilds.  Here an alignment is its run-length CIGAR plus ONE counter -- the match columns -- over
// sequences it only points to: the other counters of populate_nice_alignment (src/align.cc:274-315) follow from the
// runs (M columns - matches = mismatches; non-M runs = gaps, zero-length ones included; their lengths = gap bases), the
// match counter arrives with the CIGAR from the device (sdf_result.matches) and is corrected by comparing bases only
// where an operation cuts through columns (trims, merges).  Coordinates index the sequences `seq_a` / `seq_b` point to.
class Alignment {
 public:
  int start_a = 0, end_a = 0;
  int start_b = 0, end_b = 0;
  Cigar cigar;
  struct AlignmentError {
    int gaps, gap_bases, mismatches, matches;
  } error = {0, 0, 0, 0};

  Alignment();
  // Alignment(fa, fb): one DP over two whole strings (src/align.cc:76-88)
  Alignment(const std::string &fa, const std::string &fb, DpSession &dp);
  // a given CIGAR string over two whole strings (src/align.cc:90-105)
  Alignment(const std::string &fa, const std::string &fb, const std::string &cigar);
  // guide of refined chains + side extension (src/align.cc:107-197)
  Alignment(SeqView qstr, SeqView rstr, const std::vector<Hit> &guide, int side, DpSession &dp);
  // chain of seed anchors (src/align.cc:199-270)
  Alignment(SeqView qstr, SeqView rstr, const std::vector<Anchor> &guide, const std::vector<int> &guide_idx, DpSession &dp);

  // joins `cur`, which starts before this alignment ends, to this one (src/align.cc:505-610)
  void merge(Alignment &cur, SeqView qstr, SeqView rstr, DpSession &dp);
  void trim_front();  // keep the best-scoring suffix of the columns (src/align.cc:343-398)
  void trim_back();   // keep the best-scoring prefix (src/align.cc:400-456)

  // the alignment of columns [start, end) of this one: runs rebuilt from the remaining columns like
  // cigar_from_alignment does (src/align.cc:480-501), counters recounted; sa / sb = bases of a / b in the columns before
  // `start`, la / lb = in the kept ones (src/stats_main.cc:33-56: subhit's own counting)
  Alignment slice_columns(int start, int end, int &sa, int &la, int &sb, int &lb) const;
  // every column in order: f(column, character of a or '-', character of b or '-')
  template <typename F>
  void for_each_column(F f) const {
    int ia = start_a, ib = start_b, col = 0;
    for (auto &run : cigar)
      for (int i = 0; i < run.second; i++, col++) {
        const char ca = run.first != 'I' ? seq_a[ia++] : '-', cb = run.first != 'D' ? seq_b[ib++] : '-';
        f(col, ca, cb);
      }
  }
  const char *bases_a() const { return seq_a + start_a; }  // the bases the columns cover
  const char *bases_b() const { return seq_b + start_b; }

  std::string cigar_string() const;
  int span() const { return columns_; }  // alignment columns
  int matches() const { return error.matches; }
  int mismatches() const { return error.mismatches; }
  int gap_bases() const { return error.gap_bases; }
  int gaps() const { return error.gaps; }
  double gap_error() const;
  double mismatch_error() const;
  double total_error() const { return mismatch_error() + gap_error(); }

 private:
  void recount(int matches);  // counters and column count from the runs + the match counter
  void append(const Cigar &piece);   // run-merging concatenation (src/align.cc:469-478); adds piece.matches
  void prepend(const Cigar &piece);  // (src/align.cc:458-467)
  void fill_gap(int qfrom, int qgap, int rfrom, int rgap, DpSession &dp);
  int cut_tail(int trim, bool by_query);  // drop columns from the end until `trim` bases of one sequence are gone
  int cut_head(int trim, bool by_query);
  void normalise();                       // what rebuilding the CIGAR from columns does (src/align.cc:480-501)
  const char *seq_a = nullptr, *seq_b = nullptr;  // base x of the query / reference is seq_a[x] / seq_b[x]
  int len_a = 0, len_b = 0;                       // their lengths (side extension is clipped to them)
  int columns_ = 0;
  int matches_ = 0;
};

struct Hit {  // reference: src/hit.h:23-51
  std::shared_ptr<Sequence> query;
  int query_start = 0, query_end = 0;
  std::shared_ptr<Sequence> ref;
  int ref_start = 0, ref_end = 0;
  int jaccard = 0;
  std::string name, comment;
  Alignment aln;

  static Hit from_bed(const std::string &bed, std::string *cigar = nullptr);
  std::string to_bed(bool do_rc = true, bool with_cigar = true) const;
  bool operator<(const Hit &h) const;
  void extend(double factor, int max_extend);
};
void update_from_alignment(Hit &h);

// ---- FASTA (reference: src/fasta.{h,cc}) -----------------------------------------------------------
struct FastaIndexEntry {
  std::string name;
  int length;
  long long offset;
  int line_blen, line_len;
};
class FastaReference {
 public:
  explicit FastaReference(const std::string &filename);
  ~FastaReference();
  std::string get_sequence(const std::string &seqname, int start = 0, int *end = nullptr);
  // The same in two steps, for a caller that places the bases itself: locate() clamps `end` like get_sequence and returns the
  // bytes of the file the range spans (an upper bound of the bases: line ends are still inside; 0: nothing), extract() copies
  // the bases of such a span to `dst` and returns how many there were.
  struct Span {
    const char *src = nullptr;
    size_t bytes = 0;
  };
  Span locate(const std::string &seqname, int start, int *end) const;
  static size_t extract(const Span &s, char *dst);

 private:
  int fd_ = -1;
  void *mm_ = nullptr;
  size_t size_ = 0;
  std::map<std::string, FastaIndexEntry> index_;
};

// ---- `sedef stats generate` (reference: src/stats_main.cc:33-336), scope row f4 -----------------------
struct StatsParams {  // Globals::Stats (src/globals.cc:36-39), CLI overrides src/stats_main.cc:485-488
  int max_ok_gap = -1, min_split = 1000, min_uppercase = 100;
  double max_scaled_error = 0.5;
};
// TEST HOOK: a column walker with the oracle's signature (oracle/stats_oracle.c: sdfo_stats_columns) instead of the device
typedef int (*test_cols_fn)(const char *, int, const char *, int, const uint32_t *, int, int32_t *, char *, char *);
// Reads the BEDPE of `align generate`, writes the table of `stats generate` to `out`; the per-column counters of every
// piece come from sdf_stats_columns_batch on `device` (or from `test`).  Returns the number of lines written (header
// excluded); stats[0..2] = hits read, pieces examined, alignment columns walked.
long stats_generate(const std::string &ref_path, const std::string &bed_path, FILE *out, const StatsParams &sp,
                    test_cols_fn test, int device, long long *stats);
std::string format_double(double x);  // fmt 4.0.1 "{}" of a double, as the reference prints columns 22-25 and 35

// ---- utilities (reference: src/util.cc:33-48, src/common.h:56-99) ----------------------------------
void set_alignment_scoring(const Params &p);  // Align::MATCH and co. are process-wide (src/globals.cc:25-28)
std::vector<std::string> split(const std::string &s, char delim);
std::string rc(const std::string &s);
void rc_inplace(char *s, size_t n);  // reverse complement of bases where they lie
char align_dna(char c);
char hash_dna(char c);

// ---- chaining (reference: src/chain.cc) ------------------------------------------------------------
std::vector<Anchor> generate_anchors(const std::string &query, const std::string &ref, const Hit &orig,
                                     int kmer_size);
// returns (path, boundaries) exactly as chain_anchors (src/chain.cc:103-199)
std::pair<std::vector<int>, std::vector<std::pair<int, bool>>> chain_anchors(std::vector<Anchor> &anchors,
                                                                            const Params &p);

// test hook (tests/test_chain_oracle.py): a script of tree operations on chain_anchors' range-maximum structure
int rangemax_script(const int *pts, int n, const int *ops, int nops, int *out, int *state, int state_cap);

// ---- per-pair job: fast_align (src/chain.cc:203-268) + refine_chains (src/refine.cc:23-193), staged ----
class PairJob {
 public:
  PairJob(SeqView query, SeqView ref, const Hit &orig, const Params &p);  // (views: the sequences outlive the job)
  // Advances as far as possible.  Returns the DP requests it is waiting for (empty => finished).
  // Call again with the results of the previous return value, in the same order.
  std::vector<DpRequest> advance(const std::vector<Cigar> &results);
  bool done() const { return stage_ == DONE; }
  // anchors computed elsewhere (GPU batch): a view that stays valid until the first advance() has returned; the
  // job copies it on the thread that runs it
  void set_anchors(const Anchor *a, size_t n) {
    ext_anchors_ = a;
    ext_count_ = n;
    have_anchors_ = true;
  }
  std::vector<Hit> &hits() { return final_hits_; }

 private:
  enum Stage { START, CHAIN_ALN, PATHS, DONE };
  struct PathState;
  void stage_start(std::vector<DpRequest> &out);
  void stage_chain_finish(const std::vector<Cigar> &results);
  void plan_paths();
  void finish_paths();

  SeqView query_, ref_;
  Hit orig_;
  Params p_;
  Stage stage_ = START;
  bool have_anchors_ = false;
  std::shared_ptr<Sequence> query_ptr_, ref_ptr_;
  std::vector<Anchor> anchors_;
  const Anchor *ext_anchors_ = nullptr;
  size_t ext_count_ = 0;
  std::vector<std::vector<int>> guides_;
  std::vector<Hit> hits_;
  std::vector<std::shared_ptr<PathState>> paths_;
  std::vector<size_t> wait_counts_;  // requests issued per waiting path, in order
  std::vector<int> wait_paths_;
  std::vector<Hit> final_hits_;
  bool exact_ = false;  // the pair's sequences are plain ACGTN: device match counters are exact (DpSession)
};

// ---- `align bucket` (reference: src/align_main.cc:38-198, src/merge.cc, src/search_main.cc:93-120) -----------
struct BucketParams {  // Globals::Extend (src/globals.cc:32-34)
  double extend_ratio = 5;
  int max_extend = 15 * 1000;
  int merge_dist = 250;
};
std::vector<Hit> merge_hits(std::vector<Hit> &hits, int merge_dist);
std::vector<std::vector<std::string>> generate_translation(const std::string &ref_path);
void bucket_alignments_extern(const std::string &bed_path, int nbins, const std::string &output_dir, bool extend,
                              const std::string &reference, const BucketParams &bp, FILE *log);

// ---- the stage driver's own settings ---------------------------------------------------------------------
// (the DP library's are include/sedef_hip.h: sdf_config).  Read from the environment in ONE place -- StageSettings::from_env,
// called at the start of every run of the stage (the CLI, the sdfh_* entry points) -- and kept for that run; nothing else in
// the host code calls getenv for a setting.
struct StageSettings {
  int device = 0;            // SDF_DEVICE: the first lane's GPU
  int lanes = 0;             // SDF_LANES: super-batches in flight, each on a device context of its own (0: by the number of pairs)
  std::vector<int> devices;  // SDF_DEVICES=0,1,...: the GPUs the lanes after the first go round-robin over
  int super_batch = 0;       // SDF_SUPER_BATCH: candidate pairs per super-batch (0: by the number of pairs and lanes)
  bool gpu_anchors = true;   // SDF_GPU_ANCHORS=0: generate_anchors on the host threads
  int host_threads = 0;      // SDF_HOST_THREADS: threads of the per-pair host work (0: the CPUs the process may use, at most 64)
  double stage_ws_gib = 0;   // SDF_STAGE_WS_GIB: direction-flag workspace per lane (0: 8 GiB per process shared out)
  bool debug_timing = false; // SDF_DEBUG_TIMING: one line per phase of every super-batch
  int anchor_parts = 0;      // SDF_ANCHOR_PARTS: parts a super-batch's seed anchors are found in, each under the chaining of the one before (0: by its size -- 1, 2 or 4)
  bool resident_dp = true;   // SDF_RESIDENT_DP=0: the DP rounds cut their bases out on the host again instead of naming ranges of the characters the anchors call left in HBM
  int bucket_lanes = 2;      // SDF_BUCKET_LANES: buckets of a several-bucket run in flight, each on a device context of its own (1: one after the other)
  static StageSettings from_env();
};
const StageSettings &stage_settings();             // the current run's
void set_stage_settings(const StageSettings &s);  // (the entry points: from_env(); tests may hand over their own)

// ---- stage driver (reference: src/align_main.cc:200-337) ---------------------------------------------
struct GenerateStats {
  int lines = 0, total_written = 0;
  int64_t dp_tasks = 0, dp_cells = 0;
  int rounds = 0;
};
// Reads the bucket BED, aligns every candidate pair, writes the BEDPE lines to `out` in the reference's order.
GenerateStats generate_alignments(const std::string &ref_path, const std::string &bed_path, int kmer_size,
                                  const Params &p, DpProvider &dp, FILE *out, FILE *log, int super_batch = 8192);
// Several buckets in ONE process (the reference runs one process per bucket file, sedef.sh:187-190, and pays nothing to
// start one; a process of this build pays 0.5-0.8 s of HIP initialisation, device contexts and teardown around a stage of
// 0.2 s): every bucket's lines go to `<bucket><out_suffix>` -- the file sedef.sh redirects that bucket's stdout to --, byte
// for byte what a one-bucket run prints, and its "Finished BED" line to `log` and, with a log directory, to
// `<log_dir>/<basename>.log` (sedef.sh:195 counts those).  The providers and their lanes are set up once.
// `beds`: bucket files, or directories holding `bucket_????` files.  Returns one GenerateStats per bucket.
std::vector<std::string> expand_buckets(const std::vector<std::string> &beds);
StageHint stage_hint_many(const std::vector<std::string> &beds, int super_batch = 8192);
std::vector<GenerateStats> generate_many(const std::string &ref_path, const std::vector<std::string> &beds, int kmer_size,
                                         const Params &p, DpProvider &dp, const std::string &out_suffix,
                                         const std::string &log_dir, FILE *log, int super_batch = 8192);

}  // namespace sdfh
