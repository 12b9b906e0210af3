"""Test infrastructure: an independent Python restatement of the reference's DataTable V4 writer and reader for aggregation / group-by
intermediate results.  The product's writer is pinot_amd/csrc/host/datatable_v4.cpp; this file exists so that its bytes can be checked
against a second reading of the same Java:
  DataTableImplV4.toBytes / writeLeadingSections / serializeMetadata / serializeStringDictionary / serializeExceptions
      (pinot-common/src/main/java/org/apache/pinot/common/datatable/DataTableImplV4.java:375-391,422-558,589-606) and its ByteBuffer constructor (:133-200)
  BaseDataTableBuilder / DataTableBuilderV4 (pinot-core/src/main/java/org/apache/pinot/core/common/datatable/*.java)
  DataTableUtils.computeColumnOffsets (:41-65), DataSchema.toBytes (:118-143), AvgPair.toBytes (:57-62), ObjectType.AvgPair = 4, NULL_TYPE_VALUE = 100
PARITY UNPINNED: the reference ships no serialized DataTable fixtures and there is no JVM here to produce one, so the byte layout is
pinned only by two independent restatements agreeing (C++ and this file) and by the reader below decoding what the writer wrote."""
import struct

INT, LONG, FLOAT, DOUBLE, STRING, OBJECT = "INT", "LONG", "FLOAT", "DOUBLE", "STRING", "OBJECT"
WIDTH = {INT: 4, FLOAT: 4, STRING: 4, LONG: 8, DOUBLE: 8, OBJECT: 8}
AGG_COUNT, AGG_SUM, AGG_MIN, AGG_MAX, AGG_AVG = range(5)
AGG_NAME = {AGG_COUNT: "count", AGG_SUM: "sum", AGG_MIN: "min", AGG_MAX: "max", AGG_AVG: "avg"}
INTERMEDIATE = {AGG_COUNT: LONG, AGG_SUM: DOUBLE, AGG_MIN: DOUBLE, AGG_MAX: DOUBLE, AGG_AVG: OBJECT}
META_TYPE = {2: "l", 3: "l", 4: "l", 10: "l", 6: "i", 7: "i", 26: "i", 27: "i", 11: "s", 15: "i", 16: "l"}      # DataTable.MetadataKey ids used here
META_NAME = {2: "numDocsScanned", 3: "numEntriesScannedInFilter", 4: "numEntriesScannedPostFilter", 10: "totalDocs", 6: "numSegmentsProcessed",
             7: "numSegmentsMatched", 26: "numConsumingSegmentsProcessed", 27: "numConsumingSegmentsMatched", 11: "numGroupsLimitReached",
             15: "numResizes", 16: "resizeTimeMs"}


def _str(s):
    b = s.encode("utf-8")
    return struct.pack(">i", len(b)) + b


def _roaring(rows):
    """Portable RoaringBitmap serialization of a few small ascending ints (one array container: enough for null row ids of a test)."""
    assert rows and max(rows) < 65536 and len(rows) <= 4096
    return struct.pack("<IIHH", 12346, 1, 0, len(rows) - 1) + struct.pack("<I", 16) + b"".join(struct.pack("<H", r) for r in rows)


def encode(names, types, rows, metadata, null_handling=False, group_by=True):
    """rows: list of lists; a STRING cell is a str, an OBJECT cell an (sum, count) AvgPair or None, any other None is a null (null handling).
    An aggregation-only block (group_by=False) also lists a null OBJECT in its column's null bitmap (AggregationResultsBlock.java:119-122);
    GroupByResultsBlock.java:210 leaves OBJECT columns out."""
    offsets, at = [], 0
    for t in types:
        offsets.append(at)
        at += WIDTH[t]
    fixed, variable, dictionary = bytearray(), bytearray(), {}
    null_rows = [[] for _ in types]
    for r, row in enumerate(rows):
        buf = bytearray(at)
        for c, (t, v) in enumerate(zip(types, row)):
            o = offsets[c]
            if v is None and (t != OBJECT or not group_by):
                null_rows[c].append(r)
                v = None if t == OBJECT else ("" if t == STRING else 0)
            if t == INT:
                buf[o:o + 4] = struct.pack(">i", v)
            elif t == LONG:
                buf[o:o + 8] = struct.pack(">q", v)
            elif t == FLOAT:
                buf[o:o + 4] = struct.pack(">f", v)
            elif t == DOUBLE:
                buf[o:o + 8] = struct.pack(">d", v)
            elif t == STRING:
                buf[o:o + 4] = struct.pack(">i", dictionary.setdefault(v, len(dictionary)))
            else:
                buf[o:o + 4] = struct.pack(">i", len(variable))
                if v is None:
                    buf[o + 4:o + 8] = struct.pack(">i", 0)
                    variable += struct.pack(">i", 100)
                else:
                    buf[o + 4:o + 8] = struct.pack(">i", 16)
                    variable += struct.pack(">i", 4) + struct.pack(">dq", v[0], v[1])
        fixed += buf
    if null_handling:
        for c in range(len(types)):
            fixed += struct.pack(">i", len(variable))
            if null_rows[c]:
                b = _roaring(null_rows[c])
                fixed += struct.pack(">i", len(b))
                variable += b
            else:
                fixed += struct.pack(">i", 0)
    exceptions = struct.pack(">i", 0)
    dict_bytes = struct.pack(">i", len(dictionary)) + b"".join(_str(s) for s in dictionary)
    schema = struct.pack(">i", len(names)) + b"".join(_str(n) for n in names) + b"".join(_str(t) for t in types)
    header, at = [4, len(rows), len(names)], 13 * 4
    for section in (exceptions, dict_bytes, schema, fixed):
        header += [at, len(section)]
        at += len(section)
    header += [at, len(variable)]
    meta = struct.pack(">i", len(metadata))
    for key in sorted(metadata):
        meta += struct.pack(">i", key)
        kind, value = META_TYPE[key], metadata[key]
        meta += struct.pack(">i", int(value)) if kind == "i" else (struct.pack(">q", int(value)) if kind == "l" else _str(value))
    return struct.pack(">13i", *header) + exceptions + dict_bytes + schema + bytes(fixed) + bytes(variable) + struct.pack(">i", len(meta)) + meta


def decode(data):
    """What DataTableImplV4(ByteBuffer) + DataTableFactory read back: {names, types, rows, metadata, null_rows}."""
    version, num_rows, num_cols, exc_at, exc_len, dict_at, dict_len, schema_at, schema_len, fixed_at, fixed_len, var_at, var_len = struct.unpack_from(">13i", data, 0)
    assert version == 4 and exc_at == 52 and struct.unpack_from(">i", data, exc_at)[0] == 0

    def strings(at, count):
        out = []
        for _ in range(count):
            n = struct.unpack_from(">i", data, at)[0]
            out.append(data[at + 4:at + 4 + n].decode("utf-8"))
            at += 4 + n
        return out, at

    dictionary = strings(dict_at + 4, struct.unpack_from(">i", data, dict_at)[0])[0] if dict_len else []
    n = struct.unpack_from(">i", data, schema_at)[0]
    assert n == num_cols
    names, at = strings(schema_at + 4, n)
    types, at = strings(at, n)
    assert at == schema_at + schema_len == fixed_at
    offsets, size = [], 0
    for t in types:
        offsets.append(size)
        size += WIDTH[t]
    rows = []
    for r in range(num_rows):
        base, row = fixed_at + r * size, []
        for t, o in zip(types, offsets):
            if t == INT:
                row.append(struct.unpack_from(">i", data, base + o)[0])
            elif t == LONG:
                row.append(struct.unpack_from(">q", data, base + o)[0])
            elif t == FLOAT:
                row.append(struct.unpack_from(">f", data, base + o)[0])
            elif t == DOUBLE:
                row.append(struct.unpack_from(">d", data, base + o)[0])
            elif t == STRING:
                row.append(dictionary[struct.unpack_from(">i", data, base + o)[0]])
            else:
                pos, length = struct.unpack_from(">ii", data, base + o)
                kind = struct.unpack_from(">i", data, var_at + pos)[0]
                row.append(None if length == 0 and kind == 100 else struct.unpack_from(">dq", data, var_at + pos + 4))
                assert kind in (4, 100) and length in (0, 16)
        rows.append(row)
    null_rows = None
    if fixed_len > num_rows * size:                       # null handling: one (offset, length) pair per column behind the rows
        assert fixed_len == num_rows * size + 8 * num_cols
        null_rows = []
        for c in range(num_cols):
            pos, length = struct.unpack_from(">ii", data, fixed_at + num_rows * size + 8 * c)
            ids = []
            if length:
                cookie, containers, key, card1 = struct.unpack_from("<IIHH", data, var_at + pos)
                assert cookie == 12346 and containers == 1 and key == 0
                start = struct.unpack_from("<I", data, var_at + pos + 12)[0]
                ids = list(struct.unpack_from("<%dH" % (card1 + 1), data, var_at + pos + start))
            null_rows.append(ids)
    meta_at = var_at + var_len
    meta_len, entries = struct.unpack_from(">ii", data, meta_at)
    assert meta_at + 4 + meta_len == len(data)
    at, metadata = meta_at + 8, {}
    for _ in range(entries):
        key = struct.unpack_from(">i", data, at)[0]
        at += 4
        kind = META_TYPE[key]
        if kind == "i":
            metadata[META_NAME[key]] = struct.unpack_from(">i", data, at)[0]
            at += 4
        elif kind == "l":
            metadata[META_NAME[key]] = struct.unpack_from(">q", data, at)[0]
            at += 8
        else:
            n = struct.unpack_from(">i", data, at)[0]
            metadata[META_NAME[key]] = data[at + 4:at + 4 + n].decode("utf-8")
            at += 4 + n
    assert at == len(data)
    return {"names": names, "types": types, "rows": rows, "metadata": metadata, "null_rows": null_rows}


def results_metadata(stats, segments_processed, segments_matched, group_by=False, limit_reached=False):
    m = {10: stats[3], 2: stats[0], 3: stats[1], 4: stats[2], 6: segments_processed, 7: segments_matched, 26: 0, 27: 0}
    if group_by:
        m.update({15: 0, 16: 0})
        if limit_reached:
            m[11] = "true"
    return m
