/**
 * GroupKeyGenerator over the groups the device returned: group id k = row k of the native result, keys = the dictionary VALUES of the raw
 * group id's digits (raw id = sum dictId_j * prod_{i<j} cardinality_i, DictionaryBasedGroupKeyGenerator.java:298-338,437-445 -- the
 * same mixed-radix key the reference's ArrayBasedHolder / IntMapBasedHolder use).  Only the result-side methods are meaningful: the
 * keys were generated on the device, so generateKeysForBlock is never called.
 */
package org.apache.pinot.gpu;

import java.util.Iterator;
import org.apache.pinot.core.operator.blocks.ValueBlock;
import org.apache.pinot.core.query.aggregation.groupby.GroupKeyGenerator;
import org.apache.pinot.segment.spi.index.reader.Dictionary;


final class GpuGroupKeyGenerator implements GroupKeyGenerator {
  private final int[] _rawGroupIds;
  private final Dictionary[] _dictionaries;
  private final int[] _cardinalities;
  private final int[] _radix;
  private final int _globalUpperBound;

  /**
   * @param nullableKeys under enableNullHandling, the key columns that have null docs: their digit runs to cardinality INCLUSIVE, the last
   *                     value meaning NULL (include/pinot_gpu.h, PG_QUERY_NULL_HANDLING: the no-dictionary key generators of
   *                     DefaultGroupByExecutor.java:106-121 treat NULL as a key value of its own)
   */
  GpuGroupKeyGenerator(int[] rawGroupIds, Dictionary[] dictionaries, boolean[] nullableKeys, int globalUpperBound) {
    _rawGroupIds = rawGroupIds;
    _dictionaries = dictionaries;
    _cardinalities = new int[dictionaries.length];
    _radix = new int[dictionaries.length];
    for (int i = 0; i < dictionaries.length; i++) {
      _cardinalities[i] = dictionaries[i].length();
      _radix[i] = _cardinalities[i] + (nullableKeys[i] ? 1 : 0);
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
    return _rawGroupIds.length;
  }

  @Override
  public int getNumKeys() {
    return _rawGroupIds.length;
  }

  @Override
  public Iterator<GroupKey> getGroupKeys() {
    return new Iterator<GroupKey>() {
      private int _next = 0;
      private final GroupKey _groupKey = new GroupKey();      // reused, like the reference's iterators

      @Override
      public boolean hasNext() {
        return _next < _rawGroupIds.length;
      }

      @Override
      public GroupKey next() {
        int raw = _rawGroupIds[_next];
        Object[] keys = new Object[_dictionaries.length];
        for (int i = 0; i < _dictionaries.length; i++) {
          int digit = raw % _radix[i];
          keys[i] = digit == _cardinalities[i] ? null : _dictionaries[i].getInternal(digit);
          raw /= _radix[i];
        }
        _groupKey._groupId = _next++;
        _groupKey._keys = keys;
        return _groupKey;
      }
    };
  }
}
