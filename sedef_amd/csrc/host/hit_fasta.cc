// Hit records / BED lines (restates reference src/hit.cc) and FASTA random access (src/fasta.cc).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <tuple>

#include "sedef_host.h"

namespace sdfh {

static std::string fmt1(double v) {  // fmt "{:.1f}" == printf "%.1f"
  char buf[64];
  snprintf(buf, sizeof buf, "%.1f", v);
  return buf;
}

Hit Hit::from_bed(const std::string &bed, std::string *cigar) {  // src/hit.cc:29-63
  auto ss = split(bed, '\t');
  if (ss.size() < 10) throw std::string("BED line with fewer than 10 fields: ") + bed;
  Hit h;
  h.query = std::make_shared<Sequence>(ss[0], "", ss[8][0] != '+');
  h.ref = std::make_shared<Sequence>(ss[3], "", ss[9][0] != '+');
  h.query_start = atoi(ss[1].c_str());
  h.query_end = atoi(ss[2].c_str());
  h.ref_start = atoi(ss[4].c_str());
  h.ref_end = atoi(ss[5].c_str());
  h.name = ss[6];
  if (ss.size() >= 15) h.comment = ss[14];
  if (ss.size() >= 14) h.jaccard = atoi(ss[13].c_str());
  if (ss.size() >= 13 && cigar != nullptr) *cigar = ss[12];
  return h;
}

std::string Hit::to_bed(bool do_rc, bool with_cigar) const {  // src/hit.cc:134-196 (no translation index)
  const std::string &qn = query->name, &rn = ref->name;
  const int qs = query_start, qe = query_end;
  const int rs = do_rc && ref->is_rc ? (int)ref->seq.size() - ref_end + 1 : ref_start;
  const int re = do_rc && ref->is_rc ? (int)ref->seq.size() - ref_start + 1 : ref_end;
  std::string out;
  out += qn + "\t" + std::to_string(qs) + "\t" + std::to_string(qe) + "\t";
  out += rn + "\t" + std::to_string(rs) + "\t" + std::to_string(re) + "\t";
  out += name + "\t" + (aln.span() ? fmt1(aln.total_error()) : std::string()) + "\t";
  out += std::string(query->is_rc ? "-" : "+") + "\t" + (ref->is_rc ? "-" : "+") + "\t";
  out += std::to_string(std::max(query_end - query_start, ref_end - ref_start)) + "\t" + std::to_string(aln.span()) +
         "\t";
  if (with_cigar) out += aln.cigar_string() + "\t";
  if (aln.span()) out += "m=" + fmt1(aln.mismatch_error()) + ";g=" + fmt1(aln.gap_error());
  if (!comment.empty()) out += ";" + comment;
  return out;
}

bool Hit::operator<(const Hit &h) const {
  return std::tie(query_start, query_end, ref_start, ref_end) <
         std::tie(h.query_start, h.query_end, h.ref_start, h.ref_end);
}

void Hit::extend(double factor, int max_extend) {  // src/hit.cc:200-207
  int w = std::max(query_end - query_start, ref_end - ref_start);
  w = std::min(max_extend, int(factor * w));
  query_start = std::max(0, query_start - w);
  query_end += w;
  ref_start = std::max(0, ref_start - w);
  ref_end += w;
}

void update_from_alignment(Hit &h) {  // src/hit.cc:211-216
  h.query_start = h.aln.start_a;
  h.query_end = h.aln.end_a;
  h.ref_start = h.aln.start_b;
  h.ref_end = h.aln.end_b;
}

// ---- FASTA ----------------------------------------------------------------------------------------------
FastaReference::FastaReference(const std::string &filename) {  // src/fasta.cc:72-98
  fd_ = open(filename.c_str(), O_RDONLY);
  if (fd_ < 0) throw "Cannot open file " + filename;
  const std::string index_name = filename + ".fai";
  struct stat st_index, st_fasta;
  if (stat(index_name.c_str(), &st_index) == 0) {
    stat(filename.c_str(), &st_fasta);
    if (st_fasta.st_mtime > st_index.st_mtime)
      fprintf(stderr, "Warning: the index file is older than the FASTA file\n");
    std::ifstream fin(index_name.c_str());
    if (!fin.is_open()) throw "Index file " + index_name + " does not exist";
    std::string line;
    long long linenum = 0;
    while (std::getline(fin, line)) {  // src/fasta.cc:25-58
      ++linenum;
      auto f = split(line, '\t');
      if (f.size() != 5)
        throw "Index file " + index_name + " is malformed at line " + std::to_string(linenum);
      const std::string key = split(f[0], ' ').at(0);
      FastaIndexEntry e{f[0], atoi(f[1].c_str()), strtoll(f[2].c_str(), nullptr, 10), atoi(f[3].c_str()),
                        atoi(f[4].c_str())};
      index_.insert(std::make_pair(key, e));  // first entry of a name wins, like std::map::insert
    }
  }
  struct stat sb;
  if (fstat(fd_, &sb) == -1) throw "Cannot stat file " + filename;
  size_ = (size_t)sb.st_size;
  mm_ = mmap(nullptr, size_, PROT_READ, MAP_SHARED, fd_, 0);
  if (mm_ == MAP_FAILED) {
    mm_ = nullptr;
    throw "Cannot map file " + filename;
  }
}

FastaReference::~FastaReference() {
  if (mm_) munmap(mm_, size_);
  if (fd_ >= 0) close(fd_);
}

FastaReference::Span FastaReference::locate(const std::string &seqname, int start, int *end) const {  // src/fasta.cc:105-132
  auto it = index_.find(seqname);
  if (it == index_.end()) throw "Chromosome " + seqname + " does not exist";
  const FastaIndexEntry &entry = it->second;
  if (start < 0) start = 0;
  int length;
  if (end == nullptr || *end > entry.length) {
    length = entry.length - start;
    if (end != nullptr) *end = entry.length;
  } else {
    length = *end - start;
  }
  const int newlines_before = start > 0 ? (start - 1) / entry.line_blen : 0;
  const int newlines_by_end = (start + length - 1) / entry.line_blen;
  const int seqlen = length + (newlines_by_end - newlines_before);
  Span sp;
  if (seqlen <= 0) return sp;
  sp.src = (const char *)mm_ + entry.offset + newlines_before + start;
  sp.bytes = (size_t)seqlen;
  return sp;
}

// std::remove of '\n' then '\0' (src/fasta.cc:135-136), a line at a time
size_t FastaReference::extract(const Span &sp, char *dst) {
  const char *src = sp.src, *const stop = sp.src + sp.bytes;
  char *out = dst;
  while (src < stop) {
    const char *nl = (const char *)memchr(src, '\n', (size_t)(stop - src));
    const size_t seg = (size_t)((nl ? nl : stop) - src);
    if (memchr(src, '\0', seg)) {
      for (size_t i = 0; i < seg; i++)
        if (src[i] != '\0') *out++ = src[i];
    } else {
      memcpy(out, src, seg);
      out += seg;
    }
    src += seg + 1;
  }
  return (size_t)(out - dst);
}

std::string FastaReference::get_sequence(const std::string &seqname, int start, int *end) {  // src/fasta.cc:105-142
  const Span sp = locate(seqname, start, end);
  std::string s(sp.bytes, '\0');
  s.resize(extract(sp, &s[0]));
  return s;
}

}  // namespace sdfh
