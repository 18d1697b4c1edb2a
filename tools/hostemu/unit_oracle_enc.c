/* tools/hostemu/unit_oracle_enc.c -- the oracle's restatements of the Java entropy helpers, exported one by one for
 * tools/hostemu/check_entropy_unit.py (test infrastructure: the oracle is the checker here, as everywhere).
 *   gcc -O2 -fPIC -std=gnu11 -fno-strict-aliasing -pthread -w -shared -o tools/hostemu/libunit_oracle_enc.so tools/hostemu/unit_oracle_enc.c $(ls oracle/*.c | grep -v zstd_enc.c) */
#include "../../oracle/zstd_enc.c"

/* HuffmanCompressionTable.initialize: code lengths and values per symbol, the table's maxNumberOfBits */
void unit_ref_huf(const int32_t* counts, int32_t maxSymbol, int32_t maxBits, uint8_t* bitsOut, int16_t* valuesOut, int32_t* maxBitsOut)
{
    static huf_context ws;
    static huf_ctable t;
    memset(&t, 0, sizeof(t));
    huf_table_initialize(&t, counts, maxSymbol, maxBits, &ws);
    memcpy(bitsOut, t.numberOfBits, (size_t)maxSymbol + 1);
    memcpy(valuesOut, t.values, ((size_t)maxSymbol + 1) * 2);
    *maxBitsOut = t.maxNumberOfBits;
}
void unit_ref_norm(const int32_t* counts, int32_t total, int32_t maxSymbol, int32_t tableLog, int32_t forceSecond, int16_t* normOut)
{
    if (forceSecond) fse_normalize_counts2(normOut, tableLog, counts, total, maxSymbol);
    else fse_normalize_counts(normOut, tableLog, counts, total, maxSymbol);
}
/* returns the size, or -1 when the Java method would throw (the oracle reports that through its fail context) */
int32_t unit_ref_write(const int16_t* norm, int32_t maxSymbol, int32_t tableLog, uint8_t* out, int32_t cap)
{
    fail_ctx f;
    memset(&f, 0, sizeof(f));
    g_fail = &f;
    int32_t n = -1;
    if (setjmp(f.jb) == 0) {
        n = fse_write_normalized_counts(out, 0, cap, norm, maxSymbol, tableLog);
    }
    g_fail = NULL;
    return n;
}
/* FseCompressionTable.initialize: the state table and the two deltas per symbol (entries the method leaves alone keep the caller's fill) */
void unit_ref_fse_init(const int16_t* norm, int32_t maxSymbol, int32_t tableLog, int16_t* nextStateOut, int32_t* deltaBitsOut, int32_t* deltaFindOut)
{
    static fse_ctable t;
    for (int i = 0; i < 512; i++) t.nextState[i] = (int16_t)0x7777;
    for (int i = 0; i < 56; i++) { t.deltaNumberOfBits[i] = 0x55555555; t.deltaFindState[i] = 0x55555555; }
    fse_initialize(&t, norm, maxSymbol, tableLog);
    memcpy(nextStateOut, t.nextState, sizeof(t.nextState));
    memcpy(deltaBitsOut, t.deltaNumberOfBits, 56 * 4);
    memcpy(deltaFindOut, t.deltaFindState, 56 * 4);
}
