// jrt.h -- the few pieces of Java that tools/j2c.py's output leans on (TEST INFRASTRUCTURE: part of the recipe that builds oracle/_ref,
// never linked into the product).
//
// tools/j2c.py turns the reference's Java sources into C++ token by token; what Java provides by language or library and C++ does not is
// provided here with Java's semantics:
//   * jbyte / jshort / jint / jlong / jchar: the Java primitive widths (the unit is compiled with -fwrapv: two's-complement wrap);
//   * shifts: `a >>> b`, `a >> b`, `a << b` are emitted as `a >>JUSHR>> b`, `a >>JSHR>> b`, `a <<JSHL<< b` -- same precedence and
//     associativity as the Java operators, Java's promotion (to int unless an operand is long) and Java's masking of the count (& 31 / & 63);
//   * jarray<T>: a Java array reference (handle + length, zero-initialised, bounds-checked like the JVM: a mistranslation must not pass silently);
//   * UNSAFE: sun.misc.Unsafe's get / put / copyMemory on (base, address) pairs; a null base means an absolute address;
//   * jstring and string concatenation with numbers, the exceptions the translated sources throw, Math / Integer / Long / Short / Arrays / System / Objects;
//   * an arena that every `new` of translated code allocates from, released by the shim after each call.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <string>
#include <type_traits>
#include <utility>

typedef int8_t jbyte;
typedef int16_t jshort;
typedef int32_t jint;
typedef int64_t jlong;
typedef uint16_t jchar;
typedef float jfloat;
typedef double jdouble;

// ---- arena -------------------------------------------------------------------------------------------------------------------
namespace jrt {
struct Arena {
    struct Chunk {
        Chunk* next;
        size_t size, used;
    };
    Chunk* head = nullptr;
    void* alloc(size_t n)
    {
        n = (n + 15) & ~(size_t)15;
        if (head == nullptr || head->used + n > head->size) {
            const size_t sz = n + sizeof(Chunk) > ((size_t)4 << 20) ? n + sizeof(Chunk) : ((size_t)4 << 20);
            Chunk* c = (Chunk*)malloc(sz);
            if (c == nullptr) {
                fprintf(stderr, "jrt: out of memory\n");
                abort();
            }
            c->next = head;
            c->size = sz - sizeof(Chunk);
            c->used = 0;
            head = c;
        }
        void* p = (char*)(head + 1) + head->used;
        head->used += n;
        memset(p, 0, n);
        return p;
    }
    void release()
    {
        while (head != nullptr) {
            Chunk* n = head->next;
            free(head);
            head = n;
        }
    }
};
inline thread_local Arena arena;
inline thread_local int inCall = 0;  // > 0 while the shim runs translated code for a caller: allocations are the call's, released behind it
[[noreturn]] inline void die(const char* what)
{
    fprintf(stderr, "jrt: %s\n", what);
    abort();
}
// memory of translated code: static initialisers (class constants, tables) allocate for good, everything inside a call from the arena
inline void* alloc(size_t n)
{
    if (inCall > 0) {
        return arena.alloc(n);
    }
    void* p = calloc(n ? n : 1, 1);
    if (p == nullptr) {
        die("out of memory");
    }
    return p;
}
struct CallScope {
    CallScope() { inCall++; }
    ~CallScope()
    {
        if (--inCall == 0) {
            arena.release();
        }
    }
};
}  // namespace jrt

// every translated class derives from this: `new Foo(...)` allocates from the arena (objects are never deleted, as in Java)
struct jobject_base {
    static void* operator new(size_t n) { return jrt::alloc(n); }
    static void operator delete(void*) {}
    virtual ~jobject_base() {}
};

// ---- strings -----------------------------------------------------------------------------------------------------------------
struct jstring {
    std::string s;
    jstring() {}
    jstring(const char* c) : s(c) {}
    jstring(const std::string& c) : s(c) {}
    jint length() const { return (jint)s.size(); }
};
inline jstring operator+(const jstring& a, const jstring& b) { return jstring(a.s + b.s); }
inline jstring operator+(const jstring& a, const char* b) { return jstring(a.s + b); }
inline jstring operator+(const char* a, const jstring& b) { return jstring(a + b.s); }
template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
inline jstring operator+(const jstring& a, T b)
{
    return jstring(a.s + (std::is_same<T, bool>::value ? std::string(b ? "true" : "false") : std::to_string(b)));
}
template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
inline jstring operator+(T a, const jstring& b)
{
    return jstring((std::is_same<T, bool>::value ? std::string(a ? "true" : "false") : std::to_string(a)) + b.s);
}
struct String {
    static void fmt1(std::string& out, const std::string& f, size_t& pos) { out += f.substr(pos); pos = f.size(); }
    template <class T, class... A>
    static void fmt1(std::string& out, const std::string& f, size_t& pos, const T& v, const A&... rest)
    {
        // the next %s / %d takes v (the only conversions the reference's messages use)
        while (pos < f.size()) {
            if (f[pos] == '%' && pos + 1 < f.size() && (f[pos + 1] == 's' || f[pos + 1] == 'd')) {
                out += (jstring("") + v).s;
                pos += 2;
                fmt1(out, f, pos, rest...);
                return;
            }
            out += f[pos++];
        }
    }
    template <class... A>
    static jstring format(const jstring& f, const A&... a)
    {
        std::string out;
        size_t pos = 0;
        fmt1(out, f.s, pos, a...);
        return jstring(out);
    }
    template <class T>
    static jstring valueOf(T v)
    {
        return jstring("") + v;
    }
};

// ---- exceptions (thrown by pointer, as `throw new X(...)` reads) ---------------------------------------------------------------------
struct Throwable : jobject_base {
    jstring message;
    Throwable() {}
    Throwable(const jstring& m) : message(m) {}
    virtual jstring getMessage() { return message; }
};
struct Exception : Throwable {
    using Throwable::Throwable;
};
struct RuntimeException : Exception {
    using Exception::Exception;
};
struct IllegalArgumentException : RuntimeException {
    using RuntimeException::RuntimeException;
};
struct IllegalStateException : RuntimeException {
    using RuntimeException::RuntimeException;
};
struct UnsupportedOperationException : RuntimeException {
    using RuntimeException::RuntimeException;
};
struct IndexOutOfBoundsException : RuntimeException {
    using RuntimeException::RuntimeException;
};
struct ArrayIndexOutOfBoundsException : IndexOutOfBoundsException {
    using IndexOutOfBoundsException::IndexOutOfBoundsException;
};
struct NullPointerException : RuntimeException {
    using RuntimeException::RuntimeException;
};
struct NegativeArraySizeException : RuntimeException {
    using RuntimeException::RuntimeException;
};
struct ArithmeticException : RuntimeException {
    using RuntimeException::RuntimeException;
};
struct AssertionError : Throwable {
    using Throwable::Throwable;
};
struct IOException : Exception {
    using Exception::Exception;
};
struct EOFException : IOException {
    using IOException::IOException;
};
// io.airlift.compress.v3.MalformedInputException (M/MalformedInputException.java:16-38): (offset) / (offset, reason)
struct MalformedInputException : RuntimeException {
    jlong offset = 0;
    MalformedInputException(jlong o) : MalformedInputException(o, jstring("Malformed input")) {}
    MalformedInputException(jlong o, const jstring& reason) : RuntimeException(reason + ": offset=" + o), offset(o) {}
    jlong getOffset() { return offset; }
};

// ---- arrays ------------------------------------------------------------------------------------------------------------------
template <class T>
struct jarray {
    T* data = nullptr;
    jint length = 0;
    jarray() {}
    jarray(std::nullptr_t) {}
    jarray(std::initializer_list<T> init)
    {
        length = (jint)init.size();
        data = (T*)jrt::alloc(sizeof(T) * (init.size() ? init.size() : 1));
        jint i = 0;
        for (const T& v : init) {
            data[i++] = v;
        }
    }
    static jarray make(jlong n)
    {
        if (n < 0) {
            throw new NegativeArraySizeException(jstring("") + n);
        }
        jarray a;
        a.length = (jint)n;
        a.data = (T*)jrt::alloc(sizeof(T) * (n ? n : 1));
        return a;
    }
    T& operator[](jlong i) const
    {
        if (data == nullptr) {
            jrt::die("NullPointerException (array)");
        }
        if (i < 0 || i >= length) {  // (as the JVM: an exception the caller sees, not memory corruption)
            throw new ArrayIndexOutOfBoundsException(jstring("Index ") + i + " out of bounds for length " + length);
        }
        return data[i];
    }
    bool operator==(std::nullptr_t) const { return data == nullptr; }
    bool operator!=(std::nullptr_t) const { return data != nullptr; }
    bool operator==(const jarray& o) const { return data == o.data; }
    bool operator!=(const jarray& o) const { return data != o.data; }
    T* begin() const { return data; }
    T* end() const { return data + length; }
    jarray clone() const
    {
        jarray a = make(length);
        memcpy(a.data, data, sizeof(T) * length);
        return a;
    }
};

// java.lang.Object as the translated sources use it: the `base` of an Unsafe access -- null (absolute address) or a byte[] / short[] / int[] / long[]
struct jobject {
    void* p = nullptr;  // the array's data (Java: the array object; ARRAY_*_BASE_OFFSET below is 0)
    jobject() {}
    jobject(std::nullptr_t) {}
    template <class T>
    jobject(const jarray<T>& a) : p((void*)a.data)
    {
    }
    bool operator==(std::nullptr_t) const { return p == nullptr; }
    bool operator!=(std::nullptr_t) const { return p != nullptr; }
};

// ---- sun.misc.Unsafe ----------------------------------------------------------------------------------------------------------
// (base, address): base == null -> `address` is absolute; base == an array -> `address` is ARRAY_BASE_OFFSET + byte index.  All accesses
// are unaligned and little-endian (x86-64, as on the JVMs the reference runs on).
struct Unsafe {
    static constexpr jint ARRAY_BYTE_BASE_OFFSET = 16, ARRAY_SHORT_BASE_OFFSET = 16, ARRAY_INT_BASE_OFFSET = 16, ARRAY_LONG_BASE_OFFSET = 16;
    static uint8_t* at(const jobject& base, jlong address)
    {
        return base.p == nullptr ? (uint8_t*)(uintptr_t)address : (uint8_t*)base.p + (address - 16);
    }
    template <class T>
    static T rd(const jobject& b, jlong a)
    {
        T v;
        memcpy(&v, at(b, a), sizeof(T));
        return v;
    }
    template <class T>
    static void wr(const jobject& b, jlong a, T v)
    {
        memcpy(at(b, a), &v, sizeof(T));
    }
    jbyte getByte(const jobject& b, jlong a) const { return rd<jbyte>(b, a); }
    jshort getShort(const jobject& b, jlong a) const { return rd<jshort>(b, a); }
    jint getInt(const jobject& b, jlong a) const { return rd<jint>(b, a); }
    jlong getLong(const jobject& b, jlong a) const { return rd<jlong>(b, a); }
    void putByte(const jobject& b, jlong a, jbyte v) const { wr<jbyte>(b, a, v); }
    void putShort(const jobject& b, jlong a, jshort v) const { wr<jshort>(b, a, v); }
    void putInt(const jobject& b, jlong a, jint v) const { wr<jint>(b, a, v); }
    void putLong(const jobject& b, jlong a, jlong v) const { wr<jlong>(b, a, v); }
    jbyte getByte(jlong a) const { return rd<jbyte>(nullptr, a); }
    jshort getShort(jlong a) const { return rd<jshort>(nullptr, a); }
    jint getInt(jlong a) const { return rd<jint>(nullptr, a); }
    jlong getLong(jlong a) const { return rd<jlong>(nullptr, a); }
    void putByte(jlong a, jbyte v) const { wr<jbyte>(nullptr, a, v); }
    void putShort(jlong a, jshort v) const { wr<jshort>(nullptr, a, v); }
    void putInt(jlong a, jint v) const { wr<jint>(nullptr, a, v); }
    void putLong(jlong a, jlong v) const { wr<jlong>(nullptr, a, v); }
    void copyMemory(const jobject& sb, jlong sa, const jobject& db, jlong da, jlong n) const { memmove(at(db, da), at(sb, sa), (size_t)n); }
    void copyMemory(jlong sa, jlong da, jlong n) const { memmove(at(nullptr, da), at(nullptr, sa), (size_t)n); }
    void setMemory(const jobject& b, jlong a, jlong n, jbyte v) const { memset(at(b, a), (uint8_t)v, (size_t)n); }
    void setMemory(jlong a, jlong n, jbyte v) const { memset(at(nullptr, a), (uint8_t)v, (size_t)n); }
};
static const Unsafe UNSAFE{};
#define JASSERT(...) ((void)0)  /* the JVM runs with assertions disabled unless started with -ea */

// enums: instances with ordinal() / name(), as in Java
struct jenum_base : jobject_base {
    jint ordinal_ = 0;
    const char* name_ = "";
    jint ordinal() { return ordinal_; }
    jstring name() { return jstring(name_); }
};
template <class E>
inline E* jenum_make(E* e, jint ordinal, const char* name)
{
    e->ordinal_ = ordinal;
    e->name_ = name;
    return e;
}

// ---- shifts with Java's promotion and count masking -------------------------------------------------------------------------------
struct JUshrTag {};
struct JShrTag {};
struct JShlTag {};
static constexpr JUshrTag JUSHR{};
static constexpr JShrTag JSHR{};
static constexpr JShlTag JSHL{};
template <class T, class Tag>
struct JShiftLhs {
    T v;
};
// left operand: anything narrower than long promotes to int (Java's binary numeric promotion for shifts looks at the left operand only)
template <class T>
using jpromoted = typename std::conditional<sizeof(T) == 8, jlong, jint>::type;
template <class T, class = typename std::enable_if<std::is_integral<T>::value>::type>
constexpr JShiftLhs<jpromoted<T>, JUshrTag> operator>>(T v, JUshrTag) { return {(jpromoted<T>)v}; }
template <class T, class = typename std::enable_if<std::is_integral<T>::value>::type>
constexpr JShiftLhs<jpromoted<T>, JShrTag> operator>>(T v, JShrTag) { return {(jpromoted<T>)v}; }
template <class T, class = typename std::enable_if<std::is_integral<T>::value>::type>
constexpr JShiftLhs<jpromoted<T>, JShlTag> operator<<(T v, JShlTag) { return {(jpromoted<T>)v}; }
constexpr jint operator>>(JShiftLhs<jint, JUshrTag> l, jlong n) { return (jint)((uint32_t)l.v >> (n & 31)); }
constexpr jlong operator>>(JShiftLhs<jlong, JUshrTag> l, jlong n) { return (jlong)((uint64_t)l.v >> (n & 63)); }
constexpr jint operator>>(JShiftLhs<jint, JShrTag> l, jlong n) { return l.v >> (n & 31); }
constexpr jlong operator>>(JShiftLhs<jlong, JShrTag> l, jlong n) { return l.v >> (n & 63); }
constexpr jint operator<<(JShiftLhs<jint, JShlTag> l, jlong n) { return (jint)((uint32_t)l.v << (n & 31)); }
constexpr jlong operator<<(JShiftLhs<jlong, JShlTag> l, jlong n) { return (jlong)((uint64_t)l.v << (n & 63)); }

// ---- java.lang.Math / Integer / Long / Short / Byte, java.util.Arrays / Objects, System ----------------------------------------------
struct Math {
    static jint min(jint a, jint b) { return a < b ? a : b; }
    static jlong min(jlong a, jlong b) { return a < b ? a : b; }
    static jlong min(jint a, jlong b) { return a < b ? a : b; }
    static jlong min(jlong a, jint b) { return a < b ? a : b; }
    static jint max(jint a, jint b) { return a > b ? a : b; }
    static jlong max(jlong a, jlong b) { return a > b ? a : b; }
    static jlong max(jint a, jlong b) { return a > b ? a : b; }
    static jlong max(jlong a, jint b) { return a > b ? a : b; }
    static jint abs(jint a) { return a < 0 ? -a : a; }
    static jlong abs(jlong a) { return a < 0 ? -a : a; }
    // Math.clamp(long value, int min, int max) -> int ; (long, long, long) -> long   (Java 21)
    static jint clamp(jlong v, jint lo, jint hi)
    {
        if (lo > hi) {
            throw new IllegalArgumentException(jstring("") + lo + " > " + hi);
        }
        return (jint)(v < lo ? lo : (v > hi ? hi : v));
    }
    static jlong clamp(jlong v, jlong lo, jlong hi)
    {
        if (lo > hi) {
            throw new IllegalArgumentException(jstring("") + lo + " > " + hi);
        }
        return v < lo ? lo : (v > hi ? hi : v);
    }
    static jint toIntExact(jlong v)
    {
        if ((jint)v != v) {
            throw new ArithmeticException(jstring("integer overflow"));
        }
        return (jint)v;
    }
    static jint addExact(jint a, jint b)
    {
        jint r;
        if (__builtin_add_overflow(a, b, &r)) {
            throw new ArithmeticException(jstring("integer overflow"));
        }
        return r;
    }
    static jint multiplyExact(jint a, jint b)
    {
        jint r;
        if (__builtin_mul_overflow(a, b, &r)) {
            throw new ArithmeticException(jstring("integer overflow"));
        }
        return r;
    }
    static jint floorDiv(jint a, jint b)
    {
        jint q = a / b;
        return ((a % b != 0) && ((a < 0) != (b < 0))) ? q - 1 : q;
    }
};
struct Integer {
    static constexpr jint MAX_VALUE = 0x7FFFFFFF, MIN_VALUE = -0x7FFFFFFF - 1, BYTES = 4, SIZE = 32;
    static jint numberOfLeadingZeros(jint v) { return v == 0 ? 32 : __builtin_clz((uint32_t)v); }
    static jint numberOfTrailingZeros(jint v) { return v == 0 ? 32 : __builtin_ctz((uint32_t)v); }
    static jint highestOneBit(jint v) { return v == 0 ? 0 : (jint)(0x80000000u >> __builtin_clz((uint32_t)v)); }
    static jint bitCount(jint v) { return __builtin_popcount((uint32_t)v); }
    static jint rotateLeft(jint v, jint d) { return (jint)(((uint32_t)v << (d & 31)) | ((uint32_t)v >> ((-d) & 31))); }
    static jint rotateRight(jint v, jint d) { return (jint)(((uint32_t)v >> (d & 31)) | ((uint32_t)v << ((-d) & 31))); }
    static jint reverseBytes(jint v) { return (jint)__builtin_bswap32((uint32_t)v); }
    static jlong toUnsignedLong(jint v) { return (jlong)(uint32_t)v; }
    static jint compare(jint a, jint b) { return a < b ? -1 : (a == b ? 0 : 1); }
    static jstring toHexString(jint v)
    {
        char b[16];
        snprintf(b, sizeof(b), "%x", (unsigned)v);
        return jstring(b);
    }
    static jstring toString(jint v) { return jstring("") + v; }
};
struct Long {
    static constexpr jlong MAX_VALUE = 0x7FFFFFFFFFFFFFFFLL, MIN_VALUE = -0x7FFFFFFFFFFFFFFFLL - 1;
    static constexpr jint BYTES = 8, SIZE = 64;
    static jint numberOfLeadingZeros(jlong v) { return v == 0 ? 64 : __builtin_clzll((uint64_t)v); }
    static jint numberOfTrailingZeros(jlong v) { return v == 0 ? 64 : __builtin_ctzll((uint64_t)v); }
    static jlong rotateLeft(jlong v, jint d) { return (jlong)(((uint64_t)v << (d & 63)) | ((uint64_t)v >> ((-d) & 63))); }
    static jlong rotateRight(jlong v, jint d) { return (jlong)(((uint64_t)v >> (d & 63)) | ((uint64_t)v << ((-d) & 63))); }
    static jlong reverseBytes(jlong v) { return (jlong)__builtin_bswap64((uint64_t)v); }
    static jlong highestOneBit(jlong v) { return v == 0 ? 0 : (jlong)(0x8000000000000000ull >> __builtin_clzll((uint64_t)v)); }
    static jint bitCount(jlong v) { return __builtin_popcountll((uint64_t)v); }
    static jstring toHexString(jlong v)
    {
        char b[24];
        snprintf(b, sizeof(b), "%llx", (unsigned long long)v);
        return jstring(b);
    }
};
struct Short {
    static constexpr jshort MAX_VALUE = 0x7FFF, MIN_VALUE = -0x8000;
    static constexpr jint BYTES = 2, SIZE = 16;
    static jshort reverseBytes(jshort v) { return (jshort)__builtin_bswap16((uint16_t)v); }
    static jint toUnsignedInt(jshort v) { return (jint)(uint16_t)v; }
};
struct Byte {
    static constexpr jbyte MAX_VALUE = 0x7F, MIN_VALUE = -0x80;
    static constexpr jint BYTES = 1, SIZE = 8;
    static jint toUnsignedInt(jbyte v) { return (jint)(uint8_t)v; }
};
struct Arrays {
    template <class T, class V>
    static void fill(const jarray<T>& a, V v)
    {
        for (jint i = 0; i < a.length; i++) {
            a.data[i] = (T)v;
        }
    }
    template <class T, class V>
    static void fill(const jarray<T>& a, jint from, jint to, V v)
    {
        if (from > to) {
            throw new IllegalArgumentException(jstring("fromIndex(") + from + ") > toIndex(" + to + ")");
        }
        if (from < 0 || to > a.length) {
            throw new ArrayIndexOutOfBoundsException(jstring("Array index out of range"));
        }
        for (jint i = from; i < to; i++) {
            a.data[i] = (T)v;
        }
    }
    template <class T>
    static jarray<T> copyOf(const jarray<T>& a, jint n)
    {
        jarray<T> r = jarray<T>::make(n);
        memcpy(r.data, a.data, sizeof(T) * (size_t)(n < a.length ? n : a.length));
        return r;
    }
    template <class T>
    static jarray<T> copyOfRange(const jarray<T>& a, jint from, jint to)
    {
        jarray<T> r = jarray<T>::make(to - from);
        const jint n = (to < a.length ? to : a.length) - from;
        if (n > 0) {
            memcpy(r.data, a.data + from, sizeof(T) * (size_t)n);
        }
        return r;
    }
};
struct System {
    template <class T>
    static void arraycopy(const jarray<T>& src, jint sp, const jarray<T>& dst, jint dp, jint n)
    {
        if (sp < 0 || dp < 0 || n < 0 || sp > src.length - n || dp > dst.length - n) {
            throw new ArrayIndexOutOfBoundsException(jstring("arraycopy: last source index ") + (sp + n) + " out of bounds for length " + src.length);
        }
        memmove(dst.data + dp, src.data + sp, sizeof(T) * (size_t)n);
    }
};
struct Objects {
    template <class T>
    static T requireNonNull(T v, const jstring& what = jstring("null"))
    {
        if (v == nullptr) {
            throw new NullPointerException(what);
        }
        return v;
    }
    static jint checkFromIndexSize(jint from, jint size, jint length)
    {
        if ((length | from | size) < 0 || size > length - from) {
            throw new IndexOutOfBoundsException(jstring("Range [") + from + ", " + from + " + " + size + ") out of bounds for length " + length);
        }
        return from;
    }
    static jlong checkFromIndexSize(jlong from, jlong size, jlong length)
    {
        if ((length | from | size) < 0 || size > length - from) {
            throw new IndexOutOfBoundsException(jstring("Range out of bounds"));
        }
        return from;
    }
};

// ---- java.io.OutputStream / InputStream as the translated stream classes use them ----------------------------------------------------
struct OutputStream : jobject_base {
    virtual void write(jint b)
    {
        jarray<jbyte> one = jarray<jbyte>::make(1);
        one[0] = (jbyte)b;
        write(one, 0, 1);
    }
    virtual void write(jarray<jbyte> b) { write(b, 0, b.length); }
    virtual void write(jarray<jbyte> b, jint off, jint len)
    {
        for (jint i = 0; i < len; i++) {
            write((jint)b[off + i]);
        }
    }
    virtual void flush() {}
    virtual void close() {}
};
struct InputStream : jobject_base {
    virtual jint read() = 0;
    virtual jint read(jarray<jbyte> b) { return read(b, 0, b.length); }
    virtual jint read(jarray<jbyte> b, jint off, jint len)
    {
        if (len == 0) {
            return 0;
        }
        jint n = 0;
        while (n < len) {
            const jint c = read();
            if (c < 0) {
                break;
            }
            b[off + n++] = (jbyte)c;
        }
        return n == 0 ? -1 : n;
    }
    virtual jlong skip(jlong n)
    {
        jlong k = 0;
        while (k < n && read() >= 0) {
            k++;
        }
        return k;
    }
    virtual jint available() { return 0; }
    virtual void close() {}
};
// a sink that collects what a translated OutputStream writes (the shims' stand-in for the caller's stream)
struct ByteSink : OutputStream {
    std::string bytes;
    using OutputStream::write;
    void write(jint b) override { bytes.push_back((char)b); }
    void write(jarray<jbyte> b, jint off, jint len) override
    {
        if (off < 0 || len < 0 || off > b.length - len) {
            throw new IndexOutOfBoundsException(jstring("write out of bounds"));
        }
        bytes.append((const char*)b.data + off, (size_t)len);
    }
};
// java.io.ByteArrayInputStream(buf, offset, length) as the reference's test harness uses it (T/HadoopCodecDecompressor.java:40)
struct ByteSource : InputStream {
    const uint8_t* p;
    jlong n, pos = 0;
    ByteSource(const uint8_t* data, jlong len) : p(data), n(len) {}
    using InputStream::read;
    jint read() override { return pos < n ? (jint)p[pos++] : -1; }
    jint read(jarray<jbyte> b, jint off, jint len) override
    {
        if (off < 0 || len < 0 || len > b.length - off) {
            throw new IndexOutOfBoundsException(jstring("read out of bounds"));
        }
        if (pos >= n) {
            return -1;
        }
        const jlong avail = n - pos;
        if (len > avail) {
            len = (jint)avail;
        }
        if (len <= 0) {
            return 0;
        }
        memcpy(b.data + off, p + pos, (size_t)len);
        pos += len;
        return len;
    }
    jlong skip(jlong k) override
    {
        jlong s = n - pos < k ? n - pos : k;
        if (s < 0) {
            s = 0;
        }
        pos += s;
        return s;
    }
    jint available() override { return (jint)(n - pos); }
};
