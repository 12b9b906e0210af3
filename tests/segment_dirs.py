"""Lays Pinot segment directories out on disk for the loader tests: the reference's own v1 directories from the golden fixture,
and v1 / v3 directories built from the product's writers."""
import json
import os

import numpy as np

import helpers as H
from pinot_amd import _abi

MAGIC = bytes.fromhex("deadbeefdeafbead")      # SingleFileIndexDirectory.MAGIC_MARKER
TYPE_NAMES = {_abi.PG_TYPE_INT: "INT", _abi.PG_TYPE_LONG: "LONG", _abi.PG_TYPE_FLOAT: "FLOAT", _abi.PG_TYPE_DOUBLE: "DOUBLE"}


def reference_directories():
    return json.load(open(os.path.join(H.GOLDEN_DIR, "pinot_v1_segment_directories.json")))


def write_reference_directory(tmp_path, name):
    d = os.path.join(str(tmp_path), name)
    os.makedirs(d, exist_ok=True)
    for fname, hexdata in reference_directories()[name]["files"].items():
        with open(os.path.join(d, fname), "wb") as f:
            f.write(bytes.fromhex(hexdata))
    return d


def metadata_text(name, num_docs, columns, sorted_cols=()):
    lines = ["segment.name = %s" % name, "segment.table.name = t", "segment.total.docs = %d" % num_docs, "segment.padding.character = \\\\u0000"]
    for c in columns:
        k = "column.%s." % c.name
        has_dict = c.encoding == _abi.PG_FWD_FIXED_BIT_DICT
        lines += [k + "cardinality = %d" % c.cardinality, k + "dataType = %s" % TYPE_NAMES[c.stored_type], k + "bitsPerElement = %d" % (c.bits if has_dict else 0),
                  k + "lengthOfEachEntry = 0", k + "isSorted = %s" % ("true" if c.name in sorted_cols else "false"),
                  k + "hasDictionary = %s" % ("true" if has_dict else "false"), k + "isSingleValues = true", k + "totalDocs = %d" % num_docs]
    return "\n".join(lines) + "\n"


def sorted_forward_index(dict_ids, cardinality):
    """SortedIndexReaderImpl's file: [startDocId, endDocId] per dictId, big-endian ints."""
    out = np.zeros(2 * cardinality, dtype=">i4")
    for d in range(cardinality):
        docs = np.nonzero(dict_ids == d)[0]
        out[2 * d], out[2 * d + 1] = int(docs[0]), int(docs[-1])
    return out.view(np.uint8)


def write_v1(tmp_path, name, num_docs, columns, sorted_fwd=None):
    d = os.path.join(str(tmp_path), name)
    os.makedirs(d, exist_ok=True)
    sorted_fwd = sorted_fwd or {}
    for c in columns:
        if c.encoding == _abi.PG_FWD_FIXED_BIT_DICT:
            c.dictionary.tofile(os.path.join(d, c.name + ".dict"))
            if c.name in sorted_fwd:
                sorted_fwd[c.name].tofile(os.path.join(d, c.name + ".sv.sorted.fwd"))
            else:
                c.fwd.tofile(os.path.join(d, c.name + ".sv.unsorted.fwd"))
            if c.inverted is not None:
                c.inverted.tofile(os.path.join(d, c.name + ".bitmap.inv"))
        else:
            c.fwd.tofile(os.path.join(d, c.name + ".sv.raw.fwd"))
        if c.null_vector is not None:       # NullValueVectorCreator.seal: <column>.bitmap.nullvalue, only when some doc is null
            c.null_vector.tofile(os.path.join(d, c.name + ".bitmap.nullvalue"))
    with open(os.path.join(d, "metadata.properties"), "w") as f:
        f.write(metadata_text(name, num_docs, columns, sorted_fwd.keys()))
    return d


def write_v3(tmp_path, name, num_docs, columns):
    """<dir>/v3/{columns.psf, index_map, metadata.properties}: every index = magic marker + bytes, sizes include the marker."""
    d = os.path.join(str(tmp_path), name, "v3")
    os.makedirs(d, exist_ok=True)
    psf, index_map = bytearray(), []

    def add(column, index, data):
        index_map.append("%s.%s.startOffset = %d" % (column, index, len(psf)))
        index_map.append("%s.%s.size = %d" % (column, index, len(data) + 8))
        psf.extend(MAGIC)
        psf.extend(data)

    for c in columns:
        if c.encoding == _abi.PG_FWD_FIXED_BIT_DICT:
            add(c.name, "dictionary", c.dictionary.tobytes())
        add(c.name, "forward_index", c.fwd.tobytes())
        if c.inverted is not None:
            add(c.name, "inverted_index", c.inverted.tobytes())
        if c.null_vector is not None:
            add(c.name, "nullvalue_vector", c.null_vector.tobytes())
    with open(os.path.join(d, "columns.psf"), "wb") as f:
        f.write(bytes(psf))
    with open(os.path.join(d, "index_map"), "w") as f:
        f.write("\n".join(index_map) + "\n")
    with open(os.path.join(d, "metadata.properties"), "w") as f:
        f.write(metadata_text(name, num_docs, columns))
    return os.path.join(str(tmp_path), name)
