#!/bin/sh
# download_taxondata.sh — fetches the NCBI taxonomy dumps that make_metadata.sh needs (network access required).
# Same outcome as CuCLARK's script of this name (download_taxondata.sh:31-62):
#   ./download_taxondata.sh <directory>
# leaves nodes.dmp and merged.dmp (from taxdump.tar.gz) and nucl_accss (nucl_gb + nucl_wgs accession2taxid,
# concatenated) in <directory> and marks success with <directory>/../.taxondata.  Earlier dumps are replaced.
if [ $# -lt 1 ] || [ -z "$1" ]; then
  echo "Usage: $0 <Directory: directory to store taxonomy data> "
  echo "Note: taxonomy files already in that directory are replaced."
  exit 0
fi
TAXDIR=$1
NCBI=ftp://ftp.ncbi.nlm.nih.gov/pub/taxonomy
ACC2TAX="nucl_gb.accession2taxid nucl_wgs.accession2taxid"

mkdir -p -m 775 "$TAXDIR"
cd "$TAXDIR" || exit 1
for old in nucl_accss taxdump.tar.gz nodes.dmp merged.dmp names.dmp $ACC2TAX; do
  [ -f "$old" ] && unlink "$old"
  [ -f "$old.gz" ] && unlink "$old.gz"
done

echo "Fetching the taxonomy dumps from $NCBI ..."
for f in $ACC2TAX; do
  wget "$NCBI/accession2taxid/$f.gz" && gunzip "$f.gz"
done
wget "$NCBI/taxdump.tar.gz" && tar -zxf taxdump.tar.gz

for need in nodes.dmp merged.dmp $ACC2TAX; do
  if [ ! -s "$need" ]; then
    echo "Taxonomy download incomplete: $need is missing or empty."
    exit 1
  fi
done
cat $ACC2TAX > nucl_accss
touch ../.taxondata
echo "Taxonomy dumps ready in $TAXDIR."
