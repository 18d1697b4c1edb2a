/*
 * GoldenDump -- pins the COMPRESS side of the oracle (and of the GPU) against the real Java classes.
 *
 * Runs the reference's own Lz4JavaCompressor / SnappyJavaCompressor / ZstdJavaCompressor (the classes
 * T/benchmark/Algorithm.java:62-69 names "airlift_lz4" / "airlift_snappy" / "airlift_zstd") over every
 * file of the test corpus -- whole file in one call, and the block cuts of BASELINE configs[4]
 * (64 KiB for LZ4 / Snappy, 128 KiB for Zstd, last partial block included) -- and prints one line per stream:
 *
 *     <file> TAB <offset> TAB <length> TAB <codec> TAB <compressedLength> TAB <sha256 of the compressed bytes>
 *
 * i.e. exactly the lines of tests/golden/oracle_manifest.tsv (written by tools/make_golden.py from the C
 * oracle).  `tools/java/run_golden_dump.sh` compiles the reference with plain javac (SURVEY Appendix C),
 * runs this class and stores the output as tests/golden/java_manifest.tsv; tests/test_java_golden.py then
 * asserts java == oracle line by line (CPU suite) and java == GPU stream by stream (GPU suite).
 *
 * Needs a JDK >= 22 (the reference uses java.lang.foreign); none exists in the build container, so this
 * file has never been compiled here -- it only uses the three public classes and java.base.
 */
import io.airlift.compress.v3.Compressor;
import io.airlift.compress.v3.lz4.Lz4JavaCompressor;
import io.airlift.compress.v3.snappy.SnappyJavaCompressor;
import io.airlift.compress.v3.zstd.ZstdJavaCompressor;

import java.io.IOException;
import java.io.PrintStream;
import java.nio.charset.StandardCharsets;
import java.nio.file.Files;
import java.nio.file.Path;
import java.nio.file.Paths;
import java.security.MessageDigest;
import java.security.NoSuchAlgorithmException;
import java.util.Arrays;
import java.util.HexFormat;

public final class GoldenDump
{
    // T/benchmark/DataSet.java:28-89 minus the files that are not in the checkout (silesia/*, large/E.coli);
    // same order as tools/make_golden.py DATASET_ORDER
    private static final String[] FILES = {
            "canterbury/alice29.txt", "canterbury/asyoulik.txt", "canterbury/cp.html", "canterbury/fields.c", "canterbury/grammar.lsp",
            "canterbury/kennedy.xls", "canterbury/lcet10.txt", "canterbury/plrabn12.txt", "canterbury/ptt5", "canterbury/sum",
            "canterbury/xargs.1",
            "calgary/bib", "calgary/book1", "calgary/book2", "calgary/geo", "calgary/news", "calgary/obj1", "calgary/obj2",
            "calgary/paper1", "calgary/paper2", "calgary/paper3", "calgary/paper4", "calgary/paper5", "calgary/paper6", "calgary/pic",
            "calgary/progc", "calgary/progl", "calgary/progp", "calgary/trans",
            "artificial/a.txt", "artificial/aaa.txt", "artificial/alphabet.txt", "artificial/random.txt", "artificial/uniform_ascii.bin",
            "large/bible.txt", "large/world192.txt",
            "geo.protodata", "house.jpg", "html", "kppkn.gtb", "mapreduce-osdi-1.pdf", "urls.10K",
    };

    private GoldenDump() {}

    public static void main(String[] args)
            throws IOException, NoSuchAlgorithmException
    {
        Path testdata = Paths.get(args.length > 0 ? args[0] : "testdata");
        PrintStream out = new PrintStream(System.out, false, StandardCharsets.US_ASCII);
        String[] codecs = {"lz4", "snappy", "zstd"};
        int[] cuts = {65536, 65536, 131072};
        for (String file : FILES) {
            byte[] data = Files.readAllBytes(testdata.resolve(file));
            for (int c = 0; c < codecs.length; c++) {
                dump(out, file, codecs[c], data, 0, data.length);
                if (data.length > cuts[c]) {
                    for (int offset = 0; offset < data.length; offset += cuts[c]) {
                        dump(out, file, codecs[c], data, offset, Math.min(cuts[c], data.length - offset));
                    }
                }
            }
        }
        out.flush();
    }

    private static Compressor create(String codec)
    {
        // a fresh instance per stream: what one Compressor.compress call of a fresh codec object produces
        return switch (codec) {
            case "lz4" -> new Lz4JavaCompressor();
            case "snappy" -> new SnappyJavaCompressor();
            case "zstd" -> new ZstdJavaCompressor();
            default -> throw new IllegalArgumentException(codec);
        };
    }

    private static void dump(PrintStream out, String file, String codec, byte[] data, int offset, int length)
            throws NoSuchAlgorithmException
    {
        Compressor compressor = create(codec);
        byte[] compressed = new byte[compressor.maxCompressedLength(length)];
        int size = compressor.compress(data, offset, length, compressed, 0, compressed.length);
        MessageDigest sha = MessageDigest.getInstance("SHA-256");
        sha.update(compressed, 0, size);
        out.println(file + "\t" + offset + "\t" + length + "\t" + codec + "\t" + size + "\t" + HexFormat.of().formatHex(sha.digest()));
    }
}
