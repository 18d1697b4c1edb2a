/*
 * Licensed under the Apache License, Version 2.0 (the "License");
 * you may not use this file except in compliance with the License.
 * You may obtain a copy of the License at
 *
 *     http://www.apache.org/licenses/LICENSE-2.0
 *
 * Unless required by applicable law or agreed to in writing, software
 * distributed under the License is distributed on an "AS IS" BASIS,
 * WITHOUT WARRANTIES OR CONDITIONS OF ANY KIND, either express or implied.
 * See the License for the specific language governing permissions and
 * limitations under the License.
 */
package io.airlift.compress.v3.hip;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;

import static java.lang.foreign.ValueLayout.JAVA_INT;
import static java.lang.foreign.ValueLayout.JAVA_LONG;

/**
 * Batched entry point: many independent blocks in one call, sharded over the GPUs of a node.
 * <p>
 * Blocks never reference each other (LZ4 offsets stay inside a block, Snappy sub-blocks and Zstd frames are
 * self-contained), so the batch is cut into contiguous slices balanced by bytes, one slice per device, with no
 * collective between devices.  Each device runs its slice on its own {@link HipNative.Context}.
 */
public final class HipBatchCodec
{
    /** Result of one batch: per-block lengths, statuses (0 = ok, negative = ACHIP status) and error offsets. */
    public record Result(int[] outputLength, int[] status, long[] errorOffset) {}

    private final HipNative.Context[] contexts;

    public HipBatchCodec()
    {
        this(HipNative.deviceCount());
    }

    public HipBatchCodec(int devices)
    {
        HipNative.verifyEnabled();
        if (devices < 1 || devices > HipNative.deviceCount()) {
            throw new IllegalArgumentException("devices must be in [1, " + HipNative.deviceCount() + "]");
        }
        contexts = new HipNative.Context[devices];
        for (int i = 0; i < devices; i++) {
            contexts[i] = new HipNative.Context(i);
        }
    }

    /**
     * Runs {@code op} (HipNative.OP_*) over blocks laid out in two host segments.
     * Block i reads {@code source[sourceOffset[i] .. +sourceLength[i])} and writes
     * {@code destination[destinationOffset[i] .. +destinationCapacity[i])}.
     */
    public Result run(int op, MemorySegment source, long[] sourceOffset, int[] sourceLength,
            MemorySegment destination, long[] destinationOffset, int[] destinationCapacity)
    {
        int blocks = sourceOffset.length;
        int[] outputLength = new int[blocks];
        int[] status = new int[blocks];
        long[] errorOffset = new long[blocks];
        int[] starts = partition(sourceLength, destinationCapacity, contexts.length);

        Thread[] workers = new Thread[contexts.length];
        Throwable[] failures = new Throwable[contexts.length];
        for (int d = 0; d < contexts.length; d++) {
            int device = d;
            int first = starts[d];
            int count = starts[d + 1] - first;
            workers[d] = Thread.ofPlatform().start(() -> {
                if (count == 0) {
                    return;
                }
                try (Arena arena = Arena.ofConfined()) {
                    MemorySegment srcOff = arena.allocateFrom(JAVA_LONG, java.util.Arrays.copyOfRange(sourceOffset, first, first + count));
                    MemorySegment srcLen = arena.allocateFrom(JAVA_INT, java.util.Arrays.copyOfRange(sourceLength, first, first + count));
                    MemorySegment dstOff = arena.allocateFrom(JAVA_LONG, java.util.Arrays.copyOfRange(destinationOffset, first, first + count));
                    MemorySegment dstCap = arena.allocateFrom(JAVA_INT, java.util.Arrays.copyOfRange(destinationCapacity, first, first + count));
                    MemorySegment outLen = arena.allocate(JAVA_INT, count);
                    MemorySegment stat = arena.allocate(JAVA_INT, count);
                    MemorySegment errOff = arena.allocate(JAVA_LONG, count);
                    contexts[device].batchHost(op, source, srcOff, srcLen, destination, dstOff, dstCap, outLen, stat, errOff, count);
                    MemorySegment.copy(outLen, JAVA_INT, 0, outputLength, first, count);
                    MemorySegment.copy(stat, JAVA_INT, 0, status, first, count);
                    MemorySegment.copy(errOff, JAVA_LONG, 0, errorOffset, first, count);
                }
                catch (Throwable e) {
                    failures[device] = e;
                }
            });
        }
        for (int d = 0; d < contexts.length; d++) {
            try {
                workers[d].join();
            }
            catch (InterruptedException e) {
                Thread.currentThread().interrupt();
                throw new RuntimeException(e);
            }
            if (failures[d] != null) {
                throw new RuntimeException("device " + d + " failed", failures[d]);
            }
        }
        return new Result(outputLength, status, errorOffset);
    }

    /**
     * The same job in ONE downcall: the library makes the split (the rule of {@link #partition}) and runs every slice on its context in a
     * native host thread ({@code achip_multi_batch_host}); {@code ops} null = every item is {@code op}, otherwise one OP_* per item (a mixed
     * batch: the items of a slice are bucketed by codec inside the library).
     */
    public Result runNative(int op, int[] ops, MemorySegment source, long[] sourceOffset, int[] sourceLength,
            MemorySegment destination, long[] destinationOffset, int[] destinationCapacity)
    {
        int blocks = sourceOffset.length;
        int[] outputLength = new int[blocks];
        int[] status = new int[blocks];
        long[] errorOffset = new long[blocks];
        try (Arena arena = Arena.ofConfined()) {
            MemorySegment opsSegment = ops == null ? MemorySegment.NULL : arena.allocateFrom(JAVA_INT, ops);
            MemorySegment srcOff = arena.allocateFrom(JAVA_LONG, sourceOffset);
            MemorySegment srcLen = arena.allocateFrom(JAVA_INT, sourceLength);
            MemorySegment dstOff = arena.allocateFrom(JAVA_LONG, destinationOffset);
            MemorySegment dstCap = arena.allocateFrom(JAVA_INT, destinationCapacity);
            MemorySegment outLen = arena.allocate(JAVA_INT, Math.max(blocks, 1));
            MemorySegment stat = arena.allocate(JAVA_INT, Math.max(blocks, 1));
            MemorySegment errOff = arena.allocate(JAVA_LONG, Math.max(blocks, 1));
            HipNative.multiBatchHost(contexts, op, opsSegment, source, srcOff, srcLen, destination, dstOff, dstCap, outLen, stat, errOff, blocks, MemorySegment.NULL);
            MemorySegment.copy(outLen, JAVA_INT, 0, outputLength, 0, blocks);
            MemorySegment.copy(stat, JAVA_INT, 0, status, 0, blocks);
            MemorySegment.copy(errOff, JAVA_LONG, 0, errorOffset, 0, blocks);
        }
        return new Result(outputLength, status, errorOffset);
    }

    /** Contiguous split balanced by bytes moved (source + destination): achip_partition_blocks itself, so that the split is the library's by construction. */
    static int[] partition(int[] sourceLength, int[] destinationCapacity, int parts)
    {
        long[] weight = new long[sourceLength.length];
        for (int i = 0; i < weight.length; i++) {
            weight[i] = (long) sourceLength[i] + destinationCapacity[i];
        }
        return HipNative.partitionBlocks(weight, parts);
    }
}
