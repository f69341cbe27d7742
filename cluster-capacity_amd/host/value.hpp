// value.hpp -- the object model of the native host: Kubernetes objects as a JSON-like tree, read from JSON or from the
// YAML subset kubectl emits (block mappings / sequences, flow [] {}, quoted and plain scalars, | and > block scalars,
// multi-document streams), written as JSON or YAML.  No third-party parser: the image has none for C++.
// Scalars keep their source text: a Quantity such as 0.5, "500m" or 1e3 is interpreted by quantity.hpp, exactly like
// resource.ParseQuantity does with the JSON token (vendor/k8s.io/apimachinery/pkg/api/resource/quantity.go).
#pragma once
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <algorithm>
#include <array>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <stdexcept>
#include <thread>
#include <cstdlib>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

namespace cchost {

struct Value {
    enum Type { Null, Bool, Num, Str, Arr, Obj } t = Null;
    bool b = false;
    std::string s; // Str: the string; Num: its source text
    std::vector<Value> a;
    std::vector<std::pair<std::string, Value>> o; // insertion order kept (label / selector order matters downstream)

    static const Value &null_value() {
        static const Value v;
        return v;
    }
    static Value str(std::string x) {
        Value v;
        v.t = Str, v.s = std::move(x);
        return v;
    }
    static Value num(long long x) {
        Value v;
        v.t = Num, v.s = std::to_string(x);
        return v;
    }
    static Value boolean(bool x) {
        Value v;
        v.t = Bool, v.b = x;
        return v;
    }
    static Value array() {
        Value v;
        v.t = Arr;
        return v;
    }
    static Value object() {
        Value v;
        v.t = Obj;
        return v;
    }
    bool is_null() const { return t == Null; }
    // The reference decodes objects into typed structs: a string where a mapping belongs is a decode error there, never "absent".
    // Same here, at the accessor: null is the zero value (absent), any other kind than the one asked for is refused.
    const char *kind_name() const {
        static const char *const names[] = {"null", "a boolean", "a number", "a string", "a list", "a mapping"};
        return names[t];
    }
    [[noreturn]] void wrong_kind(const char *wanted) const { throw std::runtime_error(std::string("malformed object: expected ") + wanted + ", found " + kind_name()); }
    void want_mapping() const {
        if (t != Obj && t != Null) wrong_kind("a mapping");
    }
    // the member's value, or nullptr when absent (or when this is null): one scan where has() + [] take two
    const Value *find(const char *k, size_t len) const {
        want_mapping();
        if (t == Obj)
            for (auto &kv : o)
                if (kv.first.size() == len && std::memcmp(kv.first.data(), k, len) == 0) return &kv.second;
        return nullptr;
    }
    const Value *find(const std::string &k) const { return find(k.data(), k.size()); }
    bool has(const std::string &k) const { return has(k.data(), k.size()); }
    bool has(const char *k) const { return has(k, std::strlen(k)); }
    bool has(const char *k, size_t len) const {
        want_mapping();
        if (t != Obj) return false;
        for (auto &kv : o)
            if (kv.first.size() == len && std::memcmp(kv.first.data(), k, len) == 0) return true;
        return false;
    }
    // python's (d or {}).get(k): Null when absent (or when this is null)
    const Value &operator[](const std::string &k) const { return get(k.data(), k.size()); }
    const Value &operator[](const char *k) const { return get(k, std::strlen(k)); } // (no temporary std::string per lookup)
    const Value &get(const char *k, size_t len) const {
        want_mapping();
        if (t == Obj)
            for (auto &kv : o)
                if (kv.first.size() == len && std::memcmp(kv.first.data(), k, len) == 0) return kv.second;
        return null_value();
    }
    Value &set(const std::string &k, Value v) {
        if (t != Obj) t = Obj, o.clear();
        for (auto &kv : o)
            if (kv.first == k) return kv.second = std::move(v);
        o.emplace_back(k, std::move(v));
        return o.back().second;
    }
    // python truthiness: None, False, "", 0, [], {} are falsy
    bool truthy() const {
        switch (t) {
        case Null: return false;
        case Bool: return b;
        case Num: return !(s == "0" || s == "0.0" || s.empty());
        case Str: return !s.empty();
        case Arr: return !a.empty();
        case Obj: return !o.empty();
        }
        return false;
    }
    // text of a scalar ("" for null); booleans as YAML/JSON spell them
    std::string text() const {
        if (t == Str || t == Num) return s;
        if (t == Bool) return b ? "true" : "false";
        if (t != Null) wrong_kind("a scalar");
        return "";
    }
    // an integer field: null = absent (dflt); a number (or its text) that is not a plain decimal integer, or another kind, is refused --
    // the reference's decoder does not turn "1Gi", 5.5 or [] into an int32 either
    long long as_int(long long dflt = 0) const {
        if (t == Null) return dflt;
        if (t != Num && t != Str) wrong_kind("an integer");
        size_t i = (!s.empty() && (s[0] == '-' || s[0] == '+')) ? 1 : 0;
        bool ok = i < s.size() && s.size() - i <= 18;
        for (size_t k = i; ok && k < s.size(); k++) ok = s[k] >= '0' && s[k] <= '9';
        if (!ok) throw std::runtime_error("malformed object: expected an integer, found '" + s + "'");
        return std::stoll(s);
    }
    // ... of an int32 field (weights, maxSkew, minDomains, priority, hostPort)
    int32_t as_int32(int32_t dflt = 0) const {
        if (t == Str) wrong_kind("an integer"); // a quoted "8080" is a string to the reference's typed decoder: refused there, refused here
        const long long v = as_int(dflt);
        if (v > INT32_MAX || v < INT32_MIN) throw std::runtime_error("malformed object: " + s + " does not fit an int32 field");
        return (int32_t)v;
    }
    const std::vector<Value> &items() const {
        static const std::vector<Value> empty;
        if (t != Arr && t != Null) wrong_kind("a list");
        return t == Arr ? a : empty;
    }
    const std::vector<std::pair<std::string, Value>> &fields() const {
        static const std::vector<std::pair<std::string, Value>> empty;
        want_mapping();
        return t == Obj ? o : empty;
    }
};

// ------------------------------------------------------------------------------------------------ JSON
// Which members of a cluster object (a kubectl dump's Node / Pod / Namespace ...) the ingest reads; shared by the JSON and the YAML
// reader.  A context says where in the object a container is; member_ctx gives the context of member `k`, or -1 = skip it.
namespace prune {
enum Ctx { NoCtx, Item, Items, Metadata, Status, Spec, Containers, Container, Images };
inline bool one_of(const std::string &k, std::initializer_list<const char *> names) {
    for (const char *n : names)
        if (k == n) return true;
    return false;
}
// `kind`: of the item being parsed ("" until its "kind" member was seen); images_filtered / images_none: status.images is filtered
// by the template's image names / dropped altogether (no template names an image)
// spec.volumes of the dump's Pods: read only when a template carries volumes the volume plugins compare them with (main.cpp load_objects)
inline bool &keep_pod_volumes() {
    static bool keep = false;
    return keep;
}
inline int member_ctx(Ctx c, const std::string &k, const std::string &kind, bool images_filtered, bool images_none) {
    switch (c) {
    case Item:
        if (k == "items") return Items; // a List: its elements are objects again
        if (k == "metadata") return Metadata;
        if (k == "status") return Status;
        if (k == "spec") return Spec;
        return NoCtx;
    case Metadata:
        if (one_of(k, {"managedFields", "ownerReferences", "finalizers"})) return -1;
        // (a claim's annotations say whether its binding is complete and name its class the old way: volumes.hpp)
        if (k == "annotations" && !kind.empty() && kind != "Namespace" && kind != "PersistentVolumeClaim") return -1;
        return NoCtx;
    case Status:
        if (k == "images") return images_filtered ? (images_none ? -1 : (int)Images) : (int)NoCtx;
        return one_of(k, {"phase", "allocatable"}) ? NoCtx : -1;
    case Spec:
        if (one_of(k, {"containers", "initContainers"})) return Containers;
        if (k == "volumes") return keep_pod_volumes() ? (int)NoCtx : -1;
        if (one_of(k, {"securityContext", "imagePullSecrets", "dnsConfig", "hostAliases", "readinessGates", "tolerations", "ephemeralContainers"})) return -1;
        return NoCtx;
    case Container:
        return one_of(k, {"env", "envFrom", "volumeMounts", "volumeDevices", "livenessProbe", "readinessProbe", "startupProbe", "lifecycle", "securityContext",
                          "command", "args"}) ? -1 : NoCtx;
    default: return NoCtx;
    }
}
} // namespace prune

// `prune_cluster_objects`: the document is a kubectl dump of cluster objects (Nodes, Pods, Namespaces, ... or Lists of them)
// and only the fields the ingest reads are materialised; the rest is skipped without building it -- metadata.managedFields /
// ownerReferences / finalizers / annotations (kept for Namespaces: genpod reads them), every status field except phase /
// allocatable / images, the spec fields and container fields no scheduler plugin of this host looks at (volumes, env, probes,
// mounts, ...).  On real dumps that is most of the bytes (managedFields, last-applied-configuration, conditions, container
// statuses).  Templates (--podspec) are never read this way: they are echoed into the report as they came.
// Nesting bound of both readers (encoding/json gives up at 10000 levels; 2000 here, safe for the sanitizer builds' larger frames too -- no
// object of this path nests deeper than a few dozen):
// recursion depth is input-controlled, the stack is not.
struct DepthGuard {
    int &d;
    DepthGuard(int &depth, int limit) : d(depth) {
        if (++d > limit) {
            --d;
            throw std::runtime_error("exceeded max nesting depth " + std::to_string(limit));
        }
    }
    ~DepthGuard() { --d; }
};

class JsonParser {
  public:
    explicit JsonParser(std::string_view src, bool prune_cluster_objects = false, const std::vector<std::string> *wanted_images = nullptr)
        : s_(src), prune_(prune_cluster_objects), wanted_images_(wanted_images) {}
    Value parse_document() {
        Value v = parse_value(prune_ ? Item : NoCtx);
        skip_ws();
        if (i_ != s_.size()) fail("trailing characters");
        return v;
    }

  private:
    const std::string_view s_; // (a view: the host maps large dumps instead of copying them into a string)
    const bool prune_;
    // status.images of a Node: only entries naming one of these images are kept (ImageLocality looks at the images the template's
    // containers name, a node lists tens); nullptr = keep every entry
    const std::vector<std::string> *wanted_images_;
    size_t i_ = 0;
    using Ctx = prune::Ctx;
    static constexpr Ctx NoCtx = prune::NoCtx, Item = prune::Item, Items = prune::Items, Containers = prune::Containers, Container = prune::Container,
                         Images = prune::Images;
    std::string kind_; // of the item being parsed ("" until its "kind" member was seen)
    int member_ctx(Ctx c, const std::string &k) const { return prune::member_ctx(c, k, kind_, wanted_images_ != nullptr, wanted_images_ && wanted_images_->empty()); }
    void skip_string() { // at the opening quote
        const char *p = s_.data() + i_ + 1, *const e = s_.data() + s_.size();
        while (p < e) {
            const char c = *p++;
            if (c == '"') {
                i_ = (size_t)(p - s_.data());
                return;
            }
            if (c == '\\') p++; // whatever is escaped, it does not end the string
        }
        i_ = s_.size();
        fail("unterminated string");
    }
    // one entry of a Node's status.images, before it is parsed: can it name a wanted image?  The entry's text is searched for
    // the quoted name; an entry containing a backslash (an escape could spell the name differently) is kept to be safe.
    bool image_entry_is_wanted() {
        skip_ws();
        const size_t from = i_;
        skip_value();
        const size_t to = i_;
        i_ = from;
        if (std::memchr(s_.data() + from, '\\', to - from)) return true;
        for (const auto &w : *wanted_images_) {
            const std::string quoted = '"' + w + '"';
            if (to - from >= quoted.size() && std::search(s_.begin() + (long)from, s_.begin() + (long)to, quoted.begin(), quoted.end()) != s_.begin() + (long)to) return true;
        }
        return false;
    }
    // The "items" of a List with many elements (a cluster dump): one cheap pass finds where each element starts and ends, then
    // the elements are parsed by several threads, each with a parser of its own over its elements' text.  At the first '[' + ws.
    // -> false (nothing consumed) when the list is small or one thread is asked for (CCHOST_THREADS=1).
    bool parse_items_in_parallel(Value &v) {
        unsigned threads = std::thread::hardware_concurrency();
        if (const char *e = std::getenv("CCHOST_THREADS")) threads = (unsigned)std::atoi(e);
        threads = std::min(threads, 32u);
        size_t min_bytes = 8u << 20; // below this the threads cost more than they save (CCHOST_PARALLEL_MIN_BYTES: tests set 0)
        if (const char *e = std::getenv("CCHOST_PARALLEL_MIN_BYTES")) min_bytes = (size_t)std::atoll(e);
        if (threads < 2 || s_.size() - i_ < min_bytes) return false;
        const size_t start = i_;
        std::vector<std::pair<size_t, size_t>> spans;
        while (true) {
            skip_ws();
            const size_t from = i_;
            skip_value();
            spans.emplace_back(from, i_);
            skip_ws();
            if (i_ < s_.size() && s_[i_] == ',') {
                i_++;
                continue;
            }
            if (i_ < s_.size() && s_[i_] == ']') {
                i_++;
                break;
            }
            fail("expected ',' or ']'");
        }
        if (spans.size() < 4 * threads) {
            i_ = start;
            return false;
        }
        const size_t end = i_;
        v.a.resize(spans.size());
        std::vector<std::string> errors(threads);
        std::vector<std::thread> pool;
        // contiguous blocks of roughly equal TEXT size (objects of a dump differ in size by kind: Nodes first, then Pods)
        std::vector<size_t> cut(threads + 1, spans.size());
        cut[0] = 0;
        const size_t total = spans.back().second - spans.front().first;
        for (size_t k = 0, t = 1; k < spans.size() && t < threads; k++)
            if (spans[k].first - spans.front().first >= total * t / threads) cut[t++] = k;
        for (unsigned t = 0; t < threads; t++)
            pool.emplace_back([&, t] {
                try {
                    for (size_t k = cut[t]; k < cut[t + 1]; k++) {
                        JsonParser p(s_.substr(spans[k].first, spans[k].second - spans[k].first), prune_, wanted_images_);
                        v.a[k] = p.parse_value(Item);
                    }
                } catch (const std::exception &e) {
                    errors[t] = e.what();
                }
            });
        for (auto &th : pool) th.join();
        for (const auto &e : errors)
            if (!e.empty()) throw std::runtime_error(e);
        i_ = end;
        return true;
    }
    // Skipping a container without building it is the floor of the pruned parse (and of the pass that finds the elements'
    // boundaries for the parallel parse), so it works on 64-byte blocks: byte masks of quotes / backslashes / brackets, the
    // in-string mask as the prefix parity of the quote mask, brackets counted outside strings.  A block that contains a backslash
    // is walked byte by byte (escapes are rare outside a few long annotation strings).
    void skip_container() { // at '{' or '['
        const char *const base = s_.data(), *const e = base + s_.size();
        const char *p = base + i_;
        int depth = 0;
        bool in_string = false;
        auto bytewise = [&](const char *stop) -> bool { // -> true when the container closed (i_ set)
            while (p < stop) {
                const char c = *p++;
                if (in_string) {
                    if (c == '\\') p++; // (may step past `stop`: the escaped byte is skipped wherever it lies)
                    else if (c == '"') in_string = false;
                } else if (c == '"') in_string = true;
                else if (c == '{' || c == '[') depth++;
                else if ((c == '}' || c == ']') && --depth == 0) {
                    i_ = (size_t)(p - base);
                    return true;
                }
            }
            return false;
        };
#if defined(__SSE2__)
        const __m128i vq = _mm_set1_epi8('"'), vb = _mm_set1_epi8('\\'), vo = _mm_set1_epi8('{'), vc = _mm_set1_epi8('}'), v20 = _mm_set1_epi8(0x20);
        while (e - p >= 64) {
            uint64_t q = 0, bs = 0, op = 0, cl = 0;
            for (int k = 0; k < 4; k++) {
                const __m128i x = _mm_loadu_si128((const __m128i *)(p + 16 * k));
                const __m128i y = _mm_or_si128(x, v20); // '[' -> '{', ']' -> '}' (no other byte maps onto them)
                q |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(x, vq)) << (16 * k);
                bs |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(x, vb)) << (16 * k);
                op |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(y, vo)) << (16 * k);
                cl |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(y, vc)) << (16 * k);
            }
            if (bs) {
                if (bytewise(p + 64)) return;
                continue;
            }
            uint64_t m = q; // bit i: an odd number of quotes in bytes 0..i
            m ^= m << 1, m ^= m << 2, m ^= m << 4, m ^= m << 8, m ^= m << 16, m ^= m << 32;
            if (in_string) m = ~m; // bit i: inside a string after byte i
            in_string = (m >> 63) & 1;
            op &= ~m, cl &= ~m;
            if (!cl) depth += __builtin_popcountll(op);
            else
                for (uint64_t both = op | cl; both; both &= both - 1) {
                    const int at = __builtin_ctzll(both);
                    if ((op >> at) & 1) depth++;
                    else if (--depth == 0) {
                        i_ = (size_t)(p - base) + (size_t)at + 1;
                        return;
                    }
                }
            p += 64;
        }
#endif
        if (bytewise(e)) return;
        i_ = s_.size();
        fail("unterminated container");
    }
    void skip_value() { // no tree is built; brackets are balanced, their kinds are not cross-checked
        skip_ws();
        if (i_ >= s_.size()) fail("unexpected end");
        const char c = s_[i_];
        if (c == '"') return skip_string();
        if (c == '{' || c == '[') return skip_container();
        while (i_ < s_.size() && s_[i_] != ',' && s_[i_] != '}' && s_[i_] != ']' && s_[i_] != ' ' && s_[i_] != '\n' && s_[i_] != '\r' && s_[i_] != '\t') i_++;
    }
    [[noreturn]] void fail(const char *what) const { throw std::runtime_error("JSON: " + std::string(what) + " at offset " + std::to_string(i_)); }
    void skip_ws() {
        while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\t' || s_[i_] == '\n' || s_[i_] == '\r')) i_++;
    }
    static void put_utf8(std::string &out, unsigned cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) out += (char)(0xC0 | (cp >> 6)), out += (char)(0x80 | (cp & 0x3F));
        else if (cp < 0x10000) out += (char)(0xE0 | (cp >> 12)), out += (char)(0x80 | ((cp >> 6) & 0x3F)), out += (char)(0x80 | (cp & 0x3F));
        else out += (char)(0xF0 | (cp >> 18)), out += (char)(0x80 | ((cp >> 12) & 0x3F)), out += (char)(0x80 | ((cp >> 6) & 0x3F)), out += (char)(0x80 | (cp & 0x3F));
    }
    unsigned hex4() {
        if (i_ + 4 > s_.size()) fail("bad \\u escape");
        unsigned v = 0;
        for (int k = 0; k < 4; k++) {
            const char c = s_[i_++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= c - '0';
            else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string parse_string() {
        std::string out;
        i_++; // opening quote
        while (true) {
            // the run up to the next quote or escape in one append (kubectl dumps are almost all plain characters)
            const size_t from = i_;
            while (i_ < s_.size() && s_[i_] != '"' && s_[i_] != '\\') i_++;
            if (i_ > from) out.append(s_.data() + from, i_ - from);
            if (i_ >= s_.size()) fail("unterminated string");
            const char c = s_[i_++];
            if (c == '"') return out;
            if (i_ >= s_.size()) fail("unterminated escape");
            const char e = s_[i_++];
            switch (e) {
            case 'n': out += '\n'; break;
            case 't': out += '\t'; break;
            case 'r': out += '\r'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'u': {
                unsigned cp = hex4();
                if (cp >= 0xD800 && cp < 0xDC00 && i_ + 1 < s_.size() && s_[i_] == '\\' && s_[i_ + 1] == 'u') {
                    i_ += 2;
                    const unsigned lo = hex4();
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                }
                put_utf8(out, cp);
                break;
            }
            default: out += e; // \" \\ \/
            }
        }
    }
    int depth_ = 0;
    Value parse_value(Ctx ctx = NoCtx) {
        DepthGuard guard(depth_, 2000);
        skip_ws();
        if (i_ >= s_.size()) fail("unexpected end");
        const char c = s_[i_];
        if (c == '{') {
            Value v = Value::object();
            if (ctx == Item) kind_.clear();
            i_++;
            skip_ws();
            if (i_ < s_.size() && s_[i_] == '}') { i_++; return v; } // (not `return i_++, v`: a comma expression is no candidate for the implicit move and deep-copies v)
            while (true) {
                skip_ws();
                if (i_ >= s_.size() || s_[i_] != '"') fail("expected a key");
                std::string k = parse_string();
                skip_ws();
                if (i_ >= s_.size() || s_[i_] != ':') fail("expected ':'");
                i_++;
                const int sub = ctx == NoCtx ? (int)NoCtx : member_ctx(ctx, k);
                if (sub < 0) skip_value();
                else {
                    v.o.emplace_back(std::move(k), parse_value((Ctx)sub));
                    if (ctx == Item && v.o.back().first == "kind") kind_ = v.o.back().second.text();
                }
                skip_ws();
                if (i_ < s_.size() && s_[i_] == ',') {
                    i_++;
                    continue;
                }
                if (i_ < s_.size() && s_[i_] == '}') { i_++; return v; } // (not `return i_++, v`: a comma expression is no candidate for the implicit move and deep-copies v)
                fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            Value v = Value::array();
            i_++;
            skip_ws();
            if (i_ < s_.size() && s_[i_] == ']') { i_++; return v; } // (not `return i_++, v`: a comma expression is no candidate for the implicit move and deep-copies v)
            const Ctx elem = ctx == Items ? Item : ctx == Containers ? Container : NoCtx;
            if (ctx == Items && parse_items_in_parallel(v)) return v;
            while (true) {
                if (ctx == Images && !image_entry_is_wanted()) skip_value();
                else v.a.push_back(parse_value(elem));
                skip_ws();
                if (i_ < s_.size() && s_[i_] == ',') {
                    i_++;
                    continue;
                }
                if (i_ < s_.size() && s_[i_] == ']') { i_++; return v; } // (not `return i_++, v`: a comma expression is no candidate for the implicit move and deep-copies v)
                fail("expected ',' or ']'");
            }
        }
        if (c == '"') return Value::str(parse_string());
        if (s_.compare(i_, 4, "true") == 0) return i_ += 4, Value::boolean(true);
        if (s_.compare(i_, 5, "false") == 0) return i_ += 5, Value::boolean(false);
        if (s_.compare(i_, 4, "null") == 0) return i_ += 4, Value();
        const size_t j = i_;
        while (i_ < s_.size() && (std::isdigit((unsigned char)s_[i_]) || s_[i_] == '-' || s_[i_] == '+' || s_[i_] == '.' || s_[i_] == 'e' || s_[i_] == 'E')) i_++;
        if (j == i_) fail("unexpected character");
        Value v;
        v.t = Value::Num, v.s.assign(s_.data() + j, i_ - j);
        return v;
    }
};

inline void json_escape(std::string &out, const std::string &s) {
    out += '"';
    for (const unsigned char c : s) {
        switch (c) {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\n': out += "\\n"; break;
        case '\t': out += "\\t"; break;
        case '\r': out += "\\r"; break;
        default:
            if (c < 0x20) {
                char buf[8];
                std::snprintf(buf, sizeof buf, "\\u%04x", c);
                out += buf;
            } else
                out += (char)c;
        }
    }
    out += '"';
}

// compact JSON with ", " / ": " separators: byte-identical to python's json.dumps default
inline void to_json(std::string &out, const Value &v) {
    switch (v.t) {
    case Value::Null: out += "null"; break;
    case Value::Bool: out += v.b ? "true" : "false"; break;
    case Value::Num: out += v.s; break;
    case Value::Str: json_escape(out, v.s); break;
    case Value::Arr:
        out += '[';
        for (size_t i = 0; i < v.a.size(); i++) {
            if (i) out += ", ";
            to_json(out, v.a[i]);
        }
        out += ']';
        break;
    case Value::Obj:
        out += '{';
        for (size_t i = 0; i < v.o.size(); i++) {
            if (i) out += ", ";
            json_escape(out, v.o[i].first);
            out += ": ";
            to_json(out, v.o[i].second);
        }
        out += '}';
        break;
    }
}

// ------------------------------------------------------------------------------------------------ YAML (subset)
class YamlParser {
  public:
    // prune_cluster_objects / images: as for JsonParser (members the ingest does not read are skipped by indentation, never built;
    // status.images is dropped when no template names an image -- entries are not filtered one by one here)
    explicit YamlParser(const std::string &src, bool prune_cluster_objects = false, bool images_none = false) : prune_(prune_cluster_objects), images_none_(images_none) {
        split_lines(src);
    }
    // every document of the stream (--- separators); empty documents are dropped
    std::vector<Value> parse_stream() {
        std::vector<Value> docs;
        while (true) {
            skip_blank();
            if (pos_ >= lines_.size()) break;
            if (is_doc_marker(lines_[pos_].text)) {
                pos_++;
                continue;
            }
            next_ctx_ = prune_ ? prune::Item : prune::NoCtx;
            Value v = parse_node(lines_[pos_].indent);
            if (!v.is_null()) docs.push_back(std::move(v));
        }
        return docs;
    }

  private:
    struct Line {
        int indent;
        std::string text; // without indentation, trailing spaces and comments
        std::string raw;  // the full line (block scalars)
    };
    std::vector<Line> lines_;
    size_t pos_ = 0;

    [[noreturn]] void fail(const std::string &what) const {
        throw std::runtime_error("YAML: " + what + " near line " + std::to_string(pos_ + 1) + " (use JSON for input this subset cannot read)");
    }
    static bool is_doc_marker(const std::string &t) { return t == "---" || t.rfind("--- ", 0) == 0 || t == "..."; }
    // a quote character OPENS a quoted scalar only where a token can start (it's, q"x are plain words)
    static bool opens_quote(const std::string &t, size_t i) { return i == 0 || t[i - 1] == ' ' || t[i - 1] == '\t' || t[i - 1] == '[' || t[i - 1] == '{' || t[i - 1] == ','; }
    static std::string strip_comment(const std::string &t) {
        bool sq = false, dq = false;
        for (size_t i = 0; i < t.size(); i++) {
            const char c = t[i];
            if (sq) {
                if (c == '\'') {
                    if (i + 1 < t.size() && t[i + 1] == '\'') i++; // '' is an escaped quote
                    else sq = false;
                }
            } else if (dq) {
                if (c == '\\') i++;
                else if (c == '"') dq = false;
            } else if (c == '\'' && opens_quote(t, i)) sq = true;
            else if (c == '"' && opens_quote(t, i)) dq = true;
            else if (c == '#' && (i == 0 || t[i - 1] == ' ' || t[i - 1] == '\t')) return t.substr(0, i);
        }
        return t;
    }
    void split_lines(const std::string &src) {
        lines_.reserve((size_t)std::count(src.begin(), src.end(), '\n') + 1); // (growing a vector of two-string records by doubling cost a third of a YAML dump's read time)
        size_t i = 0;
        while (i <= src.size()) {
            size_t j = src.find('\n', i);
            if (j == std::string::npos) j = src.size();
            std::string raw = src.substr(i, j - i);
            if (!raw.empty() && raw.back() == '\r') raw.pop_back();
            Line l;
            size_t k = 0;
            while (k < raw.size() && raw[k] == ' ') k++;
            l.indent = (int)k;
            std::string t = strip_comment(raw.substr(k));
            while (!t.empty() && (t.back() == ' ' || t.back() == '\t')) t.pop_back();
            l.text = std::move(t), l.raw = std::move(raw); // (moved, not copied: this loop touches every byte of the dump several times already)
            lines_.push_back(std::move(l));
            if (j == src.size()) break;
            i = j + 1;
        }
    }
    void skip_blank() {
        while (pos_ < lines_.size() && lines_[pos_].text.empty()) pos_++;
    }
    static bool looks_like_number(const std::string &s) {
        if (s.empty()) return false;
        size_t i = (s[0] == '-' || s[0] == '+') ? 1 : 0;
        if (i >= s.size()) return false;
        bool digit = false, dot = false, exp = false;
        for (; i < s.size(); i++) {
            const char c = s[i];
            if (std::isdigit((unsigned char)c)) digit = true;
            else if (c == '.' && !dot && !exp) dot = true;
            else if ((c == 'e' || c == 'E') && digit && !exp && dot) { // YAML 1.1 (PyYAML): a float needs the dot
                exp = true;
                if (i + 1 < s.size() && (s[i + 1] == '-' || s[i + 1] == '+')) i++;
            } else
                return false;
        }
        return digit;
    }
    static Value plain_scalar(const std::string &t) {
        if (t.empty() || t == "~" || t == "null" || t == "Null" || t == "NULL") return Value();
        if (t == "true" || t == "True" || t == "TRUE") return Value::boolean(true);
        if (t == "false" || t == "False" || t == "FALSE") return Value::boolean(false);
        if (looks_like_number(t)) {
            Value v;
            v.t = Value::Num, v.s = t;
            return v;
        }
        return Value::str(t);
    }
    static void put_utf8(std::string &out, unsigned long cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) out += (char)(0xC0 | (cp >> 6)), out += (char)(0x80 | (cp & 0x3F));
        else if (cp < 0x10000) out += (char)(0xE0 | (cp >> 12)), out += (char)(0x80 | ((cp >> 6) & 0x3F)), out += (char)(0x80 | (cp & 0x3F));
        else out += (char)(0xF0 | (cp >> 18)), out += (char)(0x80 | ((cp >> 12) & 0x3F)), out += (char)(0x80 | ((cp >> 6) & 0x3F)), out += (char)(0x80 | (cp & 0x3F));
    }
    // YAML 1.1 double-quoted escapes (section 5.7): the JSON ones plus \xNN \UNNNNNNNN \0 \a \e \v \N \_ \L \P and "\ "
    static std::string unquote_double(const std::string &t, size_t &i) { // t[i] == '"'
        std::string out;
        size_t j = i + 1;
        for (; j < t.size(); j++) {
            const char c = t[j];
            if (c == '"') break;
            if (c != '\\' || j + 1 >= t.size()) {
                out += c;
                continue;
            }
            const char e = t[++j];
            auto hex = [&](int n) {
                unsigned long v = 0;
                for (int k = 0; k < n && j + 1 < t.size(); k++) {
                    const char h = t[++j];
                    v <<= 4;
                    if (h >= '0' && h <= '9') v |= (unsigned long)(h - '0');
                    else if (h >= 'a' && h <= 'f') v |= (unsigned long)(h - 'a' + 10);
                    else if (h >= 'A' && h <= 'F') v |= (unsigned long)(h - 'A' + 10);
                    else throw std::runtime_error("YAML: bad hex escape");
                }
                return v;
            };
            switch (e) {
            case 'n': out += '\n'; break;
            case 't': case '\t': out += '\t'; break;
            case 'r': out += '\r'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'v': out += '\v'; break;
            case 'a': out += '\a'; break;
            case 'e': out += '\x1b'; break;
            case '0': out += '\0'; break;
            case ' ': out += ' '; break;
            case '_': put_utf8(out, 0xA0); break;
            case 'N': put_utf8(out, 0x85); break;
            case 'L': put_utf8(out, 0x2028); break;
            case 'P': put_utf8(out, 0x2029); break;
            case 'x': put_utf8(out, hex(2)); break;
            case 'u': {
                unsigned long cp = hex(4);
                if (cp >= 0xD800 && cp < 0xDC00 && j + 2 < t.size() && t[j + 1] == '\\' && t[j + 2] == 'u') {
                    j += 2;
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (hex(4) - 0xDC00);
                }
                put_utf8(out, cp);
                break;
            }
            case 'U': put_utf8(out, hex(8)); break;
            default: out += e; // \" \\ \/
            }
        }
        if (j >= t.size()) throw std::runtime_error("YAML: unterminated double-quoted scalar");
        i = j + 1;
        return out;
    }
    static std::string unquote_single(const std::string &t, size_t &i) { // t[i] == '\''
        std::string out;
        size_t j = i + 1;
        for (; j < t.size(); j++) {
            if (t[j] == '\'') {
                if (j + 1 < t.size() && t[j + 1] == '\'') {
                    out += '\'', j++;
                    continue;
                }
                break;
            }
            out += t[j];
        }
        if (j >= t.size()) throw std::runtime_error("YAML: unterminated single-quoted scalar");
        i = j + 1;
        return out;
    }
    // flow collections and scalars inside them
    int depth_ = 0;
    Value parse_flow(const std::string &t, size_t &i) {
        DepthGuard guard(depth_, 2000);
        auto ws = [&] {
            while (i < t.size() && (t[i] == ' ' || t[i] == '\t')) i++;
        };
        ws();
        if (i >= t.size()) return Value();
        if (t[i] == '[') {
            Value v = Value::array();
            i++;
            ws();
            if (i < t.size() && t[i] == ']') { i++; return v; }
            while (true) {
                v.a.push_back(parse_flow(t, i));
                ws();
                if (i < t.size() && t[i] == ',') {
                    i++;
                    continue;
                }
                if (i < t.size() && t[i] == ']') { i++; return v; }
                fail("multi-line flow sequence");
            }
        }
        if (t[i] == '{') {
            Value v = Value::object();
            i++;
            ws();
            if (i < t.size() && t[i] == '}') { i++; return v; }
            while (true) {
                ws();
                std::string k;
                if (i < t.size() && t[i] == '"') k = unquote_double(t, i);
                else if (i < t.size() && t[i] == '\'') k = unquote_single(t, i);
                else {
                    const size_t j = i;
                    while (i < t.size() && t[i] != ':' && t[i] != ',' && t[i] != '}') i++;
                    k = t.substr(j, i - j);
                    while (!k.empty() && k.back() == ' ') k.pop_back();
                }
                ws();
                if (i >= t.size() || t[i] != ':') fail("flow mapping without ':'");
                i++;
                v.o.emplace_back(k, parse_flow(t, i));
                ws();
                if (i < t.size() && t[i] == ',') {
                    i++;
                    continue;
                }
                if (i < t.size() && t[i] == '}') { i++; return v; }
                fail("multi-line flow mapping");
            }
        }
        if (t[i] == '"') return Value::str(unquote_double(t, i));
        if (t[i] == '\'') return Value::str(unquote_single(t, i));
        const size_t j = i;
        while (i < t.size() && t[i] != ',' && t[i] != ']' && t[i] != '}') i++;
        std::string s = t.substr(j, i - j);
        while (!s.empty() && s.back() == ' ') s.pop_back();
        return plain_scalar(s);
    }
    static bool flow_balanced(const std::string &q) {
        int depth = 0;
        bool sq = false, dq = false;
        for (size_t i = 0; i < q.size(); i++) {
            const char c = q[i];
            if (sq) {
                if (c == '\'') {
                    if (i + 1 < q.size() && q[i + 1] == '\'') i++; // '' is an escaped quote
                    else sq = false;
                }
            } else if (dq) {
                if (c == '\\') i++;
                else if (c == '"') dq = false;
            } else if (c == '\'' && opens_quote(q, i)) sq = true;
            else if (c == '"' && opens_quote(q, i)) dq = true;
            else if (c == '[' || c == '{') depth++;
            else if (c == ']' || c == '}') depth--;
        }
        return depth <= 0 && !sq && !dq;
    }
    // does the quoted scalar that starts at q[0] end inside q?
    static bool quoted_terminated(const std::string &q) {
        const char quote = q[0];
        for (size_t i = 1; i < q.size(); i++) {
            if (quote == '"' && q[i] == '\\') {
                i++;
                continue;
            }
            if (q[i] == quote) {
                if (quote == '\'' && i + 1 < q.size() && q[i + 1] == '\'') {
                    i++;
                    continue;
                }
                return true;
            }
        }
        return false;
    }
    // a block scalar (| or >) whose header is on the line before pos_; parent_indent = indentation of the owning key
    Value parse_block_scalar(char style, char chomp, int parent_indent) {
        std::vector<std::string> body;
        int indent = -1;
        while (pos_ < lines_.size()) {
            const Line &l = lines_[pos_];
            const bool blank = l.raw.find_first_not_of(" \t") == std::string::npos;
            if (!blank) {
                size_t k = 0;
                while (k < l.raw.size() && l.raw[k] == ' ') k++;
                if ((int)k <= parent_indent) break;
                if (indent < 0) indent = (int)k;
                if ((int)k < indent) break;
            }
            body.push_back(blank ? "" : l.raw.substr((size_t)indent));
            pos_++;
        }
        while (!body.empty() && body.back().empty()) body.pop_back();
        std::string out;
        for (size_t i = 0; i < body.size(); i++) {
            out += body[i];
            if (i + 1 < body.size()) out += (style == '|' || body[i].empty() || body[i + 1].empty()) ? "\n" : " ";
        }
        if (chomp != '-' && !body.empty()) out += '\n';
        return Value::str(out);
    }
    // the value that follows "key:" or "- " on the same line (rest), or on the following lines
    Value parse_inline_or_nested(const std::string &rest, int owner_indent, bool owner_is_seq_item) {
        if (!rest.empty()) {
            if (rest[0] == '|' || rest[0] == '>') {
                const char chomp = rest.size() > 1 && (rest[1] == '-' || rest[1] == '+') ? rest[1] : ' ';
                return parse_block_scalar(rest[0], chomp, owner_indent);
            }
            if (rest[0] == '"' || rest[0] == '\'') {
                // a quoted scalar may be folded over several lines: a line break is a space, an empty line a newline
                std::string q = rest;
                while (!quoted_terminated(q) && pos_ < lines_.size()) {
                    const std::string &raw = lines_[pos_].raw;
                    const size_t b = raw.find_first_not_of(" \t");
                    while (!q.empty() && (q.back() == ' ' || q.back() == '\t')) q.pop_back();
                    if (b == std::string::npos) q += "\x01"; // empty line (placeholder: resolved below)
                    else {
                        if (q.back() == '\x01' || (q.back() == '\\' && q[0] == '"')) { // after a newline / an escaped line break: no space
                            if (q.back() == '\\') q.pop_back();
                        } else
                            q += ' ';
                        size_t e = raw.size();
                        while (e > b && (raw[e - 1] == ' ' || raw[e - 1] == '\t' || raw[e - 1] == '\r')) e--;
                        q += raw.substr(b, e - b);
                    }
                    pos_++;
                }
                size_t i = 0;
                Value v = parse_flow(q, i);
                if (v.t == Value::Str)
                    for (auto &c : v.s)
                        if (c == '\x01') c = '\n';
                return v;
            }
            if (rest[0] == '[' || rest[0] == '{') {
                std::string q = rest; // a flow collection may continue on the following lines until its brackets balance
                while (!flow_balanced(q) && pos_ < lines_.size()) {
                    const std::string &raw = lines_[pos_].raw;
                    const size_t b = raw.find_first_not_of(" \t");
                    if (b != std::string::npos) q += " " + strip_comment(raw.substr(b));
                    pos_++;
                }
                size_t i = 0;
                Value v = parse_flow(q, i);
                return v;
            }
            if (rest[0] == '&' || rest[0] == '*' || rest[0] == '!') fail("anchors / aliases / tags");
            // plain scalar, possibly continued on more-indented lines (kubectl folds long strings)
            std::string s = rest;
            while (pos_ < lines_.size()) {
                const Line &l = lines_[pos_];
                if (l.text.empty()) break;
                if (l.indent <= owner_indent) break;
                if (l.text[0] == '-' && (l.text.size() == 1 || l.text[1] == ' ')) break;
                if (find_key_colon(l.text) != std::string::npos) break;
                s += " " + l.text;
                pos_++;
            }
            return plain_scalar(s);
        }
        skip_blank();
        if (pos_ >= lines_.size()) return Value();
        const Line &l = lines_[pos_];
        if (is_doc_marker(l.text)) return Value();
        const bool seq_here = l.text[0] == '-' && (l.text.size() == 1 || l.text[1] == ' ');
        // a sequence may sit at the SAME indentation as its key ("containers:\n- name: x"), a mapping must be deeper
        if (l.indent > owner_indent || (seq_here && l.indent == owner_indent && !owner_is_seq_item)) return parse_node(l.indent);
        return Value();
    }
    // position of the ':' that ends a block-mapping key, npos if the line is not "key: ..."
    static size_t find_key_colon(const std::string &t) {
        if (t.empty() || t[0] == '[' || t[0] == '{') return std::string::npos;
        size_t i = 0;
        if (t[0] == '"') { // a quoted key: skip to its closing quote
            for (i = 1; i < t.size() && t[i] != '"'; i++)
                if (t[i] == '\\') i++;
            i++;
        } else if (t[0] == '\'') {
            for (i = 1; i < t.size(); i++)
                if (t[i] == '\'') {
                    if (i + 1 < t.size() && t[i + 1] == '\'') i++;
                    else break;
                }
            i++;
        }
        for (; i < t.size(); i++)
            if (t[i] == ':' && (i + 1 == t.size() || t[i + 1] == ' ')) return i;
        return std::string::npos;
    }
    static std::string key_text(std::string k) {
        while (!k.empty() && k.back() == ' ') k.pop_back();
        if (k.size() >= 2 && k.front() == '"') {
            size_t i = 0;
            return unquote_double(k, i);
        }
        if (k.size() >= 2 && k.front() == '\'') {
            size_t i = 0;
            return unquote_single(k, i);
        }
        return k;
    }
    const bool prune_, images_none_;
    prune::Ctx next_ctx_ = prune::NoCtx; // the context of the next container parse_mapping / parse_sequence opens (consumed there)
    std::string kind_;
    prune::Ctx take_ctx() {
        const prune::Ctx c = next_ctx_;
        next_ctx_ = prune::NoCtx;
        return c;
    }
    // a skipped member: everything more indented than its key, plus -- for a key with nothing after the colon -- the sequence items
    // that sit at the key's own indentation ("key:\n- a\n- b")
    void skip_member(int key_indent, bool rest_empty) {
        while (pos_ < lines_.size()) {
            const Line &l = lines_[pos_];
            if (l.text.empty()) {
                pos_++;
                continue;
            }
            if (is_doc_marker(l.text)) break;
            const bool item_here = rest_empty && l.indent == key_indent && l.text[0] == '-' && (l.text.size() == 1 || l.text[1] == ' ');
            if (l.indent > key_indent || item_here) pos_++;
            else break;
        }
    }
    Value parse_node(int indent) {
        DepthGuard guard(depth_, 2000);
        skip_blank();
        if (pos_ >= lines_.size()) return Value();
        const std::string &first = lines_[pos_].text;
        if (first[0] == '-' && (first.size() == 1 || first[1] == ' ')) return parse_sequence(indent);
        if (find_key_colon(first) != std::string::npos) return parse_mapping(indent);
        // a bare scalar document
        std::string t = first;
        pos_++;
        size_t i = 0;
        if (t[0] == '[' || t[0] == '{' || t[0] == '"' || t[0] == '\'') return parse_flow(t, i);
        return plain_scalar(t);
    }
    Value parse_sequence(int indent) {
        Value v = Value::array();
        const prune::Ctx ctx = take_ctx();
        const prune::Ctx elem = ctx == prune::Items ? prune::Item : ctx == prune::Containers ? prune::Container : prune::NoCtx;
        while (true) {
            skip_blank();
            if (pos_ >= lines_.size()) break;
            Line &l = lines_[pos_];
            if (l.indent != indent || is_doc_marker(l.text)) break;
            if (!(l.text[0] == '-' && (l.text.size() == 1 || l.text[1] == ' '))) break;
            next_ctx_ = elem;
            std::string rest = l.text.size() > 1 ? l.text.substr(2) : "";
            size_t lead = 0;
            while (lead < rest.size() && rest[lead] == ' ') lead++;
            rest = rest.substr(lead);
            const int item_indent = indent + 2 + (int)lead; // where the item's content starts
            if (rest == "-" || rest.rfind("- ", 0) == 0) { // "- - x": the item is itself a sequence that starts on this line
                l.text = rest, l.indent = item_indent;
                v.a.push_back(parse_sequence(item_indent));
                continue;
            }
            if (!rest.empty() && find_key_colon(rest) != std::string::npos && rest[0] != '"' && rest[0] != '\'' && rest[0] != '[' && rest[0] != '{') {
                // "- key: value": the item is a mapping whose first key sits on the dash line
                l.text = rest, l.indent = item_indent;
                v.a.push_back(parse_mapping(item_indent));
            } else if (!rest.empty() && (rest[0] == '"' || rest[0] == '\'') && find_key_colon(rest) != std::string::npos) {
                l.text = rest, l.indent = item_indent;
                v.a.push_back(parse_mapping(item_indent));
            } else {
                pos_++;
                v.a.push_back(parse_inline_or_nested(rest, indent, true));
            }
        }
        next_ctx_ = prune::NoCtx;
        return v;
    }
    Value parse_mapping(int indent) {
        Value v = Value::object();
        const prune::Ctx ctx = take_ctx();
        if (ctx == prune::Item) kind_.clear();
        while (true) {
            skip_blank();
            if (pos_ >= lines_.size()) break;
            const Line &l = lines_[pos_];
            if (l.indent != indent || is_doc_marker(l.text)) break;
            const size_t c = find_key_colon(l.text);
            if (c == std::string::npos) break;
            const std::string key = key_text(l.text.substr(0, c));
            std::string rest = c + 1 < l.text.size() ? l.text.substr(c + 1) : "";
            size_t lead = 0;
            while (lead < rest.size() && rest[lead] == ' ') lead++;
            rest = rest.substr(lead);
            pos_++;
            const int sub = ctx == prune::NoCtx ? (int)prune::NoCtx : prune::member_ctx(ctx, key, kind_, true, images_none_);
            if (sub < 0) {
                skip_member(indent, rest.empty());
                continue;
            }
            next_ctx_ = sub == prune::Images ? prune::NoCtx : (prune::Ctx)sub; // (image entries are not filtered here)
            v.o.emplace_back(key, parse_inline_or_nested(rest, indent, false));
            next_ctx_ = prune::NoCtx; // (a scalar value opened no container)
            if (ctx == prune::Item && key == "kind") kind_ = v.o.back().second.text();
        }
        return v;
    }
};

inline bool yaml_needs_quotes(const std::string &s) {
    if (s.empty()) return true;
    static const char *special[] = {"true", "false", "null", "True", "False", "Null", "yes", "no", "on", "off", "~", "Yes", "No", "y", "n"};
    for (const char *w : special)
        if (s == w) return true;
    if (std::isdigit((unsigned char)s[0]) || s[0] == '-' || s[0] == '+' || s[0] == '.') {
        bool numeric = true;
        for (const char c : s)
            if (!(std::isdigit((unsigned char)c) || c == '.' || c == '-' || c == '+' || c == 'e' || c == 'E' || c == '_' || c == ':')) numeric = false;
        if (numeric) return true;
    }
    if (std::string("!&*-?|>'\"%@`#{}[],").find(s[0]) != std::string::npos && !(s[0] == '-' && s.size() > 1 && s[1] != ' ')) return true;
    if (s.back() == ' ' || s.front() == ' ') return true;
    for (size_t i = 0; i < s.size(); i++) {
        const char c = s[i];
        if (c == '\n' || c == '\t' || c == '"' || (unsigned char)c < 0x20) return true;
        if (c == ':' && (i + 1 == s.size() || s[i + 1] == ' ')) return true;
        if (c == '#' && i > 0 && s[i - 1] == ' ') return true;
    }
    return false;
}

// block-style YAML in the shape python's yaml.safe_dump(sort_keys=False) produces for these reviews
// the same tree with every mapping's members in key order: what a YAML document looks like after sigs.k8s.io/yaml marshalled it
// through JSON (the reference's -o yaml, report.go:296-303, and its PrintPod)
inline Value sorted_keys(Value v) {
    for (auto &x : v.a) x = sorted_keys(std::move(x));
    for (auto &kv : v.o) kv.second = sorted_keys(std::move(kv.second));
    std::stable_sort(v.o.begin(), v.o.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    return v;
}

inline void to_yaml(std::string &out, const Value &v, int indent = 0, bool in_seq_item = false) {
    const std::string pad((size_t)indent, ' ');
    auto scalar = [&](const Value &x) {
        switch (x.t) {
        case Value::Null: out += "null"; break;
        case Value::Bool: out += x.b ? "true" : "false"; break;
        case Value::Num: out += x.s; break;
        case Value::Str:
            if (yaml_needs_quotes(x.s)) {
                out += '\'';
                for (const char c : x.s) {
                    if (c == '\'') out += "''";
                    else out += c;
                }
                out += '\'';
            } else
                out += x.s;
            break;
        default: break;
        }
    };
    if (v.t == Value::Obj) {
        if (v.o.empty()) {
            out += "{}\n";
            return;
        }
        bool first = true;
        for (auto &kv : v.o) {
            if (!(first && in_seq_item)) out += pad;
            first = false;
            { Value k = Value::str(kv.first); scalar(k); }
            out += ':';
            const Value &c = kv.second;
            if (c.t == Value::Obj && !c.o.empty()) out += '\n', to_yaml(out, c, indent + 2);
            else if (c.t == Value::Arr && !c.a.empty()) out += '\n', to_yaml(out, c, indent);
            else if (c.t == Value::Obj) out += " {}\n";
            else if (c.t == Value::Arr) out += " []\n";
            else out += ' ', scalar(c), out += '\n';
        }
        return;
    }
    if (v.t == Value::Arr) {
        if (v.a.empty()) {
            out += "[]\n";
            return;
        }
        for (auto &c : v.a) {
            out += pad + "- ";
            if (c.t == Value::Obj && !c.o.empty()) to_yaml(out, c, indent + 2, true);
            else if (c.t == Value::Arr && !c.a.empty()) to_yaml(out, c, indent + 2, true);
            else if (c.t == Value::Obj) out += "{}\n";
            else if (c.t == Value::Arr) out += "[]\n";
            else scalar(c), out += '\n';
        }
        return;
    }
    scalar(v);
    out += '\n';
}

// a file of objects: JSON (first non-blank character is { or [) or YAML stream
inline std::vector<Value> parse_documents(std::string_view text, bool prune_cluster_objects = false, const std::vector<std::string> *wanted_images = nullptr) {
    size_t i = 0;
    while (i < text.size() && std::isspace((unsigned char)text[i])) i++;
    std::string json_error;
    if (i < text.size() && (text[i] == '{' || text[i] == '[')) {
        try {
            JsonParser p(text, prune_cluster_objects, wanted_images);
            std::vector<Value> one;
            one.push_back(p.parse_document()); // (moved: a braced list would copy the whole document)
            return one;
        } catch (const std::exception &e) { // a YAML document in flow style?
            if (text.size() > (32u << 20)) throw; // (not at this size: a broken multi-hundred-MB JSON dump should say where it is broken)
            json_error = e.what();
        }
    }
    try {
        YamlParser y{std::string(text), prune_cluster_objects, wanted_images != nullptr && wanted_images->empty()};
        return y.parse_stream();
    } catch (const std::exception &e) {
        if (json_error.empty()) throw;
        throw std::runtime_error(json_error + " (and not YAML either: " + e.what() + ")"); // it began like JSON: say where THAT reading broke
    }
}

} // namespace cchost
