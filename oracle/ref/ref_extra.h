// ref_extra.h -- hand-written stand-ins for reference classes that are plumbing, not codec arithmetic (TEST INFRASTRUCTURE, part of the
// oracle/_ref recipe).  Everything here is a few lines of argument checking whose Java original uses generics / varargs (outside the
// subset tools/j2c.py translates); each cites the Java lines it stands in for.
#pragma once
#include "jrt.h"

// M/snappy/SnappyInternalUtils.java:27-43 (copied there from Guava's Preconditions): checkNotNull / checkArgument with a format string
struct SnappyInternalUtils {
    template <class T, class... A>
    static T checkNotNull(T reference, const jstring& errorMessageTemplate, const A&... args)
    {
        if (reference == nullptr) {
            throw new NullPointerException(String::format(errorMessageTemplate, args...));
        }
        return reference;
    }
    template <class... A>
    static void checkArgument(bool expression, const jstring& errorMessageTemplate, const A&... args)
    {
        if (!expression) {
            throw new IllegalArgumentException(String::format(errorMessageTemplate, args...));
        }
    }
    // :45-51
    static void checkPositionIndexes(jint start, jint end, jint size)
    {
        if (start < 0 || end < start || end > size) {
            throw new IndexOutOfBoundsException(jstring("bad position indexes"));
        }
    }
};
