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
package io.airlift.compress.v3.xxhash;

import io.airlift.compress.v3.hip.HipNative;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;
import java.lang.foreign.ValueLayout;
import java.lang.invoke.MethodHandle;

/**
 * One-shot and batched XXH64 / XXH32 on an AMD GPU (MI355X, gfx950) through {@code libaircompressor_hip.so}:
 * the GPU siblings of {@code XxHash64Hasher.hash(MemorySegment, long)} and {@code XxHash32Hasher.hash(MemorySegment, int)}.
 * <p>
 * A single host segment is staged through the context's pinned buffer (PCIe-bound; the CPU hashers are the better
 * choice for that).  The batched form hashes many device-resident buffers per call -- block / content checksums of
 * containers whose blocks already live in HBM -- at the HBM read rate.
 * <p>
 * Binding (added to {@code HipNative.MethodHandles}):
 * <pre>
 * &#64;NativeSignature(name = "achip_xxhash64", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, long.class, long.class, MemorySegment.class})
 * &#64;NativeSignature(name = "achip_xxhash32", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, long.class, int.class, MemorySegment.class})
 * &#64;NativeSignature(name = "achip_xxhash64_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, long.class, MemorySegment.class, int.class})
 * &#64;NativeSignature(name = "achip_xxhash32_batch", returnType = int.class, argumentTypes = {MemorySegment.class, MemorySegment.class, MemorySegment.class, MemorySegment.class, int.class, MemorySegment.class, int.class})
 * </pre>
 */
public final class XxHashHip
{
    private final HipNative.Context context;

    public XxHashHip(int device)
    {
        HipNative.verifyEnabled();
        this.context = new HipNative.Context(device);
    }

    /** {@code XxHash64Hasher.hash(input, seed)} on the GPU. */
    public long hash64(MemorySegment input, long seed)
    {
        try (Arena arena = Arena.ofConfined()) {
            MemorySegment out = arena.allocate(ValueLayout.JAVA_LONG);
            int status = invoke(HipNative.xxhash64(), context.address(), input, input.byteSize(), seed, out);
            HipNative.throwIfError(status, 0);
            return out.get(ValueLayout.JAVA_LONG, 0);
        }
    }

    /** {@code XxHash32Hasher.hash(input, seed)} on the GPU. */
    public int hash32(MemorySegment input, int seed)
    {
        try (Arena arena = Arena.ofConfined()) {
            MemorySegment out = arena.allocate(ValueLayout.JAVA_INT);
            int status = invoke(HipNative.xxhash32(), context.address(), input, input.byteSize(), seed, out);
            HipNative.throwIfError(status, 0);
            return out.get(ValueLayout.JAVA_INT, 0);
        }
    }

    /**
     * Hashes {@code count} device-resident buffers: buffer i is {@code base + offsets[i]}, {@code lengths[i]} bytes;
     * all segments are device memory obtained from {@code HipNative.Context.deviceAlloc}; asynchronous on the context's stream.
     */
    public void hash64Batch(MemorySegment base, MemorySegment offsets, MemorySegment lengths, long seed, MemorySegment hashes, int count)
    {
        int status = invoke(HipNative.xxhash64Batch(), context.address(), base, offsets, lengths, seed, hashes, count);
        HipNative.throwIfError(status, 0);
    }

    private static int invoke(MethodHandle handle, Object... arguments)
    {
        try {
            return (int) handle.invokeWithArguments(arguments);
        }
        catch (Throwable t) {
            throw new AssertionError("should not reach here", t);
        }
    }
}
