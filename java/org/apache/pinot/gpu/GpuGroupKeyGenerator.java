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
  private final int _globalUpperBound;

  GpuGroupKeyGenerator(int[] rawGroupIds, Dictionary[] dictionaries, int globalUpperBound) {
    _rawGroupIds = rawGroupIds;
    _dictionaries = dictionaries;
    _cardinalities = new int[dictionaries.length];
    for (int i = 0; i < dictionaries.length; i++) {
      _cardinalities[i] = dictionaries[i].length();
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
          keys[i] = _dictionaries[i].getInternal(raw % _cardinalities[i]);
          raw /= _cardinalities[i];
        }
        _groupKey._groupId = _next++;
        _groupKey._keys = keys;
        return _groupKey;
      }
    };
  }
}
