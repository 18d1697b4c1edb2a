/*
 * GoldenStreamDump -- pins the oracle's restatement of ZstdOutputStream (oracle/zstd_enc.c orc_zstd_stream_compress; the device writer
 * behind achip_zstdstream_compress) against the real class.
 *
 * For every file of the test corpus, and for the whole corpus as ONE stream ("*": ~14 MB -- the stream then flushes chunks before close(),
 * slides its window, and, BlockCompressionState.windowBaseOffset staying where it was, writes blocks without a single match after every
 * slide: DESIGN.md 10 row 3), it drives io.airlift.compress.v3.zstd.ZstdOutputStream the way T/HadoopCodecCompressor.java:57-72 does --
 * one write(buffer, 0, n), then close() -- and prints
 *
 *     <file> TAB 0 TAB <length> TAB zstdstream TAB <compressedLength> TAB <sha256 of what reached the sink>
 *
 * i.e. the lines of tests/golden/oracle_stream_manifest.tsv (tools/make_golden.py).  tools/java/run_golden_dump.sh stores the output as
 * tests/golden/java_stream_manifest.tsv; tests/test_java_golden.py asserts java == oracle line by line.  Never compiled here (no JDK).
 */
import io.airlift.compress.v3.zstd.ZstdOutputStream;

import java.io.ByteArrayOutputStream;
import java.io.IOException;
import java.io.PrintStream;
import java.nio.charset.StandardCharsets;
import java.nio.file.Files;
import java.nio.file.Path;
import java.nio.file.Paths;
import java.security.MessageDigest;
import java.security.NoSuchAlgorithmException;
import java.util.HexFormat;

public final class GoldenStreamDump
{
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

    private GoldenStreamDump() {}

    public static void main(String[] args)
            throws IOException, NoSuchAlgorithmException
    {
        Path testdata = Paths.get(args.length > 0 ? args[0] : "testdata");
        PrintStream out = new PrintStream(System.out, false, StandardCharsets.US_ASCII);
        ByteArrayOutputStream all = new ByteArrayOutputStream();
        for (String file : FILES) {
            byte[] data = Files.readAllBytes(testdata.resolve(file));
            dump(out, file, data);
            all.write(data);
        }
        dump(out, "*", all.toByteArray());
        out.flush();
    }

    private static void dump(PrintStream out, String file, byte[] data)
            throws IOException, NoSuchAlgorithmException
    {
        ByteArrayOutputStream sink = new ByteArrayOutputStream();
        ZstdOutputStream stream = new ZstdOutputStream(sink);
        stream.write(data, 0, data.length);
        stream.close();
        byte[] compressed = sink.toByteArray();
        MessageDigest sha = MessageDigest.getInstance("SHA-256");
        sha.update(compressed, 0, compressed.length);
        out.println(file + "\t0\t" + data.length + "\tzstdstream\t" + compressed.length + "\t" + HexFormat.of().formatHex(sha.digest()));
    }
}
