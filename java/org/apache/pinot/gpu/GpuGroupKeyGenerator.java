/**
 * GroupKeyGenerator over the groups the device returned: group id k = row k of the native result, keys = the dictionary VALUES of the
 * key's dictIds, which the device hands back as they are (pg_result.group_key_dict_ids: one dictId per group-by column and group; for a
 * raw INT / LONG column the entry is value - min and the key is the value, as the reference's no-dictionary key generators have it) --
 * whichever RawKeyHolder the key space calls for in the reference: Array / IntMap (raw key an int), LongMap (a long) or ArrayMap
 * (beyond a long), DictionaryBasedGroupKeyGenerator.java:150-184.  Only the result-side methods are meaningful: the keys were
 * generated on the device, so generateKeysForBlock is never called.
 */
package org.apache.pinot.gpu;

import java.util.Iterator;
import org.apache.pinot.core.operator.blocks.ValueBlock;
import org.apache.pinot.core.query.aggregation.groupby.GroupKeyGenerator;
import org.apache.pinot.segment.spi.index.reader.Dictionary;


import org.apache.pinot.spi.data.FieldSpec;


final class GpuGroupKeyGenerator implements GroupKeyGenerator {
  private final int _numGroups;
  private final int[] _keyDictIds;            // [group * columns + column]
  private final Dictionary[] _dictionaries;   // null for a raw (no-dictionary) key column
  private final int[] _nullEntries;           // the entry that means NULL: cardinality, or a raw column's max - min + 1
  private final long[] _bases;                // raw key columns: key value = base + entry (pg_group_key_info)
  private final boolean[] _longKeys;          // raw key columns: stored type LONG (the key is a Long, else an Integer)
  private final long[][] _rankValues;         // raw key columns keyed through a rank image (keyInfo isOffset 2): the values behind the entries
  private final FieldSpec.DataType[] _storedTypes;
  private final int _globalUpperBound;

  /**
   * @param keyDictIds the dictId of every group's key in every group-by column, row-major by group.  Under enableNullHandling the digit of a
   *                   key column that has null docs runs to cardinality INCLUSIVE, the last value meaning NULL (include/pinot_gpu.h,
   *                   PG_QUERY_NULL_HANDLING: the no-dictionary key generators of DefaultGroupByExecutor.java:106-121 treat NULL as a
   *                   key value of its own)
   * @param keyInfo    per key column PinotGpuNative.groupKeyInfo {base, isOffset, nullEntry}: a raw INT / LONG column (dictionaries[i] ==
   *                   null) is keyed by VALUE like NoDictionarySingleColumnGroupKeyGenerator.java:100-113 does -- base + entry
   * @param longKeys   per key column: the stored type is LONG
   */
  GpuGroupKeyGenerator(int numGroups, int[] keyDictIds, Dictionary[] dictionaries, long[][] keyInfo, boolean[] longKeys, long[][] rankValues,
      FieldSpec.DataType[] storedTypes, int globalUpperBound) {
    if (keyDictIds.length != numGroups * dictionaries.length) {
      throw new IllegalStateException("native result: " + keyDictIds.length + " key dictIds for " + numGroups + " groups of " + dictionaries.length + " columns");
    }
    _numGroups = numGroups;
    _keyDictIds = keyDictIds;
    _dictionaries = dictionaries;
    _nullEntries = new int[dictionaries.length];
    _bases = new long[dictionaries.length];
    _longKeys = longKeys;
    _rankValues = rankValues;
    _storedTypes = storedTypes;
    for (int i = 0; i < dictionaries.length; i++) {
      _bases[i] = keyInfo[i][0];
      _nullEntries[i] = (int) keyInfo[i][2];
      if ((dictionaries[i] == null) != (keyInfo[i][1] != 0)) {
        throw new IllegalStateException("group-by column " + i + ": the segment and the device disagree about its dictionary");
      }
    }
    _globalUpperBound = globalUpperBound;
  }

  @Override
  public int getGlobalGroupKeyUpperBound() {
    return _globalUpperBound;
  }

  @Override
  public void generateKeysForBlock(ValueBlock valueBlock, int[] groupKeys) {
    throw new UnsupportedOperationException("group keys are generated on the device");
  }

  @Override
  public void generateKeysForBlock(ValueBlock valueBlock, int[][] groupKeys) {
    throw new UnsupportedOperationException("group keys are generated on the device");
  }

  @Override
  public int getCurrentGroupKeyUpperBound() {
    return _numGroups;
  }

  @Override
  public int getNumKeys() {
    return _numGroups;
  }

  @Override
  public Iterator<GroupKey> getGroupKeys() {
    return new Iterator<GroupKey>() {
      private int _next = 0;
      private final GroupKey _groupKey = new GroupKey();      // reused, like the reference's iterators

      @Override
      public boolean hasNext() {
        return _next < _numGroups;
      }

      @Override
      public GroupKey next() {
        int columns = _dictionaries.length;
        Object[] keys = new Object[columns];
        for (int i = 0; i < columns; i++) {
          int entry = _keyDictIds[_next * columns + i];
          if (entry == _nullEntries[i]) {
            keys[i] = null;
          } else if (_dictionaries[i] != null) {
            keys[i] = _dictionaries[i].getInternal(entry);
          } else if (_rankValues[i] != null) {
            // keyed by value through the device-built dictionary: the map key types of NoDictionarySingleColumnGroupKeyGenerator.java:100-135
            long bits = _rankValues[i][entry];
            switch (_storedTypes[i]) {
              case INT:
                keys[i] = (int) bits;
                break;
              case LONG:
                keys[i] = bits;
                break;
              case FLOAT:
                keys[i] = (float) Double.longBitsToDouble(bits);
                break;
              default:
                keys[i] = Double.longBitsToDouble(bits);
                break;
            }
          } else if (_longKeys[i]) {
            keys[i] = _bases[i] + entry;               // Long
          } else {
            keys[i] = (int) (_bases[i] + entry);       // Integer
          }
        }
        _groupKey._groupId = _next++;
        _groupKey._keys = keys;
        return _groupKey;
      }
    };
  }
}
