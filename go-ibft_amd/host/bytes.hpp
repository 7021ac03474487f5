// bytes.hpp — the byte-string type of the host mirror.
//
// Product host code.  A []byte of the Go side: value semantics, comparisons, hashing — with two properties the hot path
// needs and std::string does not have:
//   * 24 bytes of inline storage: a 20-byte address (every store key, every quorum set member) never touches the heap;
//   * BORROWED views: a decoded message's fields (From, Signature, hashes, seals, raw proposals) point into the buffer
//     the message was decoded from, which the message keeps alive (IbftMessage::backing) — decoding a COMMIT costs one
//     allocation (the message object) instead of five, decoding the 29 241 messages nested in a round change's
//     certificates one each instead of five each.
// A COPY of a bytes always owns its data (so a key copied into a map, a seal handed to the application, outlive the
// buffer); only bytes::view() creates a borrowed one, and only moves preserve it.  Mutating a view detaches it first.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <random>
#include <functional>
#include <string>
#include <string_view>

namespace ibft {

class bytes {
 public:
  static constexpr uint32_t kInline = 24;
  using value_type = char;
  using const_iterator = const char *;
  using iterator = const char *;

  bytes() noexcept : p_(inl_), n_(0), cap_(kInline) {}
  bytes(const char *s, size_t n) { init_copy(s, n); }
  bytes(const char *cstr) { init_copy(cstr, cstr ? strlen(cstr) : 0); }
  bytes(const std::string &s) { init_copy(s.data(), s.size()); }
  bytes(size_t n, char c) {
    init_copy(nullptr, 0);
    resize(n, c);
  }
  bytes(const bytes &o) { init_copy(o.p_, o.n_); }
  bytes(bytes &&o) noexcept { steal(o); }
  ~bytes() { release(); }
  bytes &operator=(const bytes &o) {
    if (this != &o) assign(o.p_, o.n_);
    return *this;
  }
  bytes &operator=(bytes &&o) noexcept {
    if (this != &o) {
      release();
      steal(o);
    }
    return *this;
  }
  // borrowed: [p, p + n) must outlive this object and every object it is MOVED into
  static bytes view(const char *p, size_t n) noexcept {
    bytes b;
    b.p_ = const_cast<char *>(p);
    b.n_ = (uint32_t)n;
    b.cap_ = 0;
    return b;
  }
  bool is_view() const noexcept { return cap_ == 0; }

  const char *data() const noexcept { return p_; }
  size_t size() const noexcept { return n_; }
  size_t length() const noexcept { return n_; }
  bool empty() const noexcept { return n_ == 0; }
  const char *begin() const noexcept { return p_; }
  const char *end() const noexcept { return p_ + n_; }
  char operator[](size_t i) const noexcept { return p_[i]; }
  char back() const noexcept { return p_[n_ - 1]; }
  operator std::string_view() const noexcept { return std::string_view(p_, n_); }
  std::string str() const { return std::string(p_, n_); }

  void clear() noexcept {
    if (is_view()) {
      p_ = inl_;
      cap_ = kInline;
    }
    n_ = 0;
  }
  bytes &assign(const char *s, size_t n) {
    if (is_view() || n > cap_) {
      // s may alias our own buffer only when it IS a view of it; a fresh buffer keeps that safe
      char *np = n <= kInline ? inl_ : static_cast<char *>(malloc(n));
      if (n) memmove(np, s, n);
      if (!is_view() && cap_ > kInline) free(p_);
      p_ = np;
      cap_ = n <= kInline ? kInline : (uint32_t)n;
    } else if (n) {
      memmove(p_, s, n);
    }
    n_ = (uint32_t)n;
    return *this;
  }
  bytes &assign(const bytes &o) { return assign(o.p_, o.n_); }
  void reserve(size_t want) {
    if (is_view()) detach(want);
    if (want <= cap_) return;
    size_t nc = cap_ * 2 > want ? cap_ * 2 : want;
    char *np = static_cast<char *>(malloc(nc));
    if (n_) memcpy(np, p_, n_);
    if (cap_ > kInline) free(p_);
    p_ = np;
    cap_ = (uint32_t)nc;
  }
  void resize(size_t n, char c = 0) {
    reserve(n);
    if (n > n_) memset(p_ + n_, c, n - n_);
    n_ = (uint32_t)n;
  }
  bytes &append(const char *s, size_t n) {
    if (!n) return *this;
    reserve((size_t)n_ + n);
    memcpy(p_ + n_, s, n);
    n_ += (uint32_t)n;
    return *this;
  }
  bytes &append(const bytes &o) { return append(o.p_, o.n_); }
  bytes &operator+=(const bytes &o) { return append(o.p_, o.n_); }
  bytes &operator+=(char c) {
    push_back(c);
    return *this;
  }
  void push_back(char c) {
    reserve((size_t)n_ + 1);
    p_[n_++] = c;
  }
  bytes substr(size_t pos, size_t n = (size_t)-1) const {
    if (pos > n_) pos = n_;
    if (n > n_ - pos) n = n_ - pos;
    return bytes(p_ + pos, n);
  }
  int compare(const bytes &o) const noexcept {
    const size_t m = n_ < o.n_ ? n_ : o.n_;
    const int c = m ? memcmp(p_, o.p_, m) : 0;
    return c ? c : (n_ < o.n_ ? -1 : n_ > o.n_ ? 1 : 0);
  }
  friend bool operator==(const bytes &a, const bytes &b) noexcept { return a.n_ == b.n_ && (a.n_ == 0 || memcmp(a.p_, b.p_, a.n_) == 0); }
  friend bool operator!=(const bytes &a, const bytes &b) noexcept { return !(a == b); }
  friend bool operator<(const bytes &a, const bytes &b) noexcept { return a.compare(b) < 0; }
  friend bytes operator+(const bytes &a, const bytes &b) {
    bytes o;
    o.reserve(a.size() + b.size());
    o.append(a);
    o.append(b);
    return o;
  }

 private:
  void init_copy(const char *s, size_t n) {
    if (n <= kInline) {
      p_ = inl_;
      cap_ = kInline;
    } else {
      p_ = static_cast<char *>(malloc(n));
      cap_ = (uint32_t)n;
    }
    if (n) memcpy(p_, s, n);
    n_ = (uint32_t)n;
  }
  void detach(size_t want) {  // a view becomes an owner (with room for `want` bytes)
    const char *src = p_;
    const size_t n = n_, cap = want > n ? want : n;
    if (cap <= kInline) {
      p_ = inl_;
      cap_ = kInline;
    } else {
      p_ = static_cast<char *>(malloc(cap));
      cap_ = (uint32_t)cap;
    }
    if (n) memcpy(p_, src, n);
  }
  void release() noexcept {
    if (cap_ > kInline) free(p_);
  }
  void steal(bytes &o) noexcept {
    n_ = o.n_;
    cap_ = o.cap_;
    if (o.cap_ == kInline && !o.is_view()) {  // inline owner: copy the bytes
      p_ = inl_;
      if (n_) memcpy(inl_, o.inl_, n_);
    } else {  // heap owner or view: take the pointer
      p_ = o.p_;
    }
    o.p_ = o.inl_;
    o.n_ = 0;
    o.cap_ = kInline;
  }
  char *p_;
  uint32_t n_, cap_;  // cap_: 0 = borrowed view, kInline = inline storage, more = heap capacity
  char inl_[kInline];
};

// hash of a short key (an address): 8 bytes at a time, multiply-fold, keyed per process — the keys are attacker-chosen
// bytes (the From of any message, of any message nested in a certificate), and an unkeyed hash would let colliding keys be
// prepared offline to turn every open-addressing table here into a linear list
inline uint64_t hash_seed() noexcept {
  static const uint64_t seed = [] {
    std::random_device rd;
    return ((uint64_t)rd() << 32 | rd()) | 1;
  }();
  return seed;
}
inline uint64_t hash_key(const char *p, size_t n) noexcept {
  uint64_t h = ((uint64_t)n * 0x9E3779B97F4A7C15ull) ^ hash_seed();
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    h = (h ^ w) * 0xFF51AFD7ED558CCDull;
    h ^= h >> 32;
    p += 8;
    n -= 8;
  }
  if (n) {
    uint64_t w = 0;
    memcpy(&w, p, n);
    h = (h ^ w) * 0xC4CEB9FE1A85EC53ull;
    h ^= h >> 29;
  }
  return h;
}
struct bytes_hash {
  size_t operator()(const bytes &b) const noexcept { return (size_t)hash_key(b.data(), b.size()); }
};

}  // namespace ibft

namespace std {
template <>
struct hash<ibft::bytes> {
  size_t operator()(const ibft::bytes &b) const noexcept { return ibft::bytes_hash()(b); }
};
}  // namespace std
