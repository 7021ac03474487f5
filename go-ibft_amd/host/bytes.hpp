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

// Hash of a short key (an address): SipHash-1-3 under a 128-bit per-process key.  The keys are attacker-chosen bytes (the
// From of any message, of any message nested in a certificate) and they index every open-addressing table here (SenderMap,
// LeanView, the validator seats, the certificate walks' sender sets): a multiply / xor-shift hash has differentials that do
// not depend on its seed (round-3 advice: keys (w1, w2) and (w1 ^ 2^63, w2 ^ (2^63 | 2^31)) collided under every seed),
// so colliding keys could be prepared offline whatever the seed and the tables pushed towards linear probing.  SipHash is a
// PRF: without the key no collision can be prepared.  A 20-byte address costs three compression rounds + three final ones.
struct SipKey {
  uint64_t k0, k1;
};
inline const SipKey &hash_seed() noexcept {
  static const SipKey key = [] {
    std::random_device rd;
    return SipKey{(uint64_t)rd() << 32 | rd(), (uint64_t)rd() << 32 | rd()};
  }();
  return key;
}
inline uint64_t siphash13(const char *p, size_t n, const SipKey &key) noexcept {
  uint64_t v0 = key.k0 ^ 0x736f6d6570736575ull, v1 = key.k1 ^ 0x646f72616e646f6dull, v2 = key.k0 ^ 0x6c7967656e657261ull,
           v3 = key.k1 ^ 0x7465646279746573ull;
  auto rotl = [](uint64_t x, int b) { return (x << b) | (x >> (64 - b)); };
  auto round = [&]() {
    v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
    v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
    v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
    v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
  };
  uint64_t b = (uint64_t)n << 56;
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    v3 ^= w;
    round();
    v0 ^= w;
    p += 8;
    n -= 8;
  }
  if (n) {
    uint64_t w = 0;
    memcpy(&w, p, n);
    b |= w;
  }
  v3 ^= b;
  round();
  v0 ^= b;
  v2 ^= 0xff;
  round();
  round();
  round();
  return v0 ^ v1 ^ v2 ^ v3;
}
inline uint64_t hash_key(const char *p, size_t n) noexcept { return siphash13(p, n, hash_seed()); }
struct sv_hash {  // the same keyed hash for std:: containers of address views
  size_t operator()(std::string_view s) const noexcept { return (size_t)hash_key(s.data(), s.size()); }
};
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
